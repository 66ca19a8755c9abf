"""The reference-held published outputs at the frame-index boundary (evaluation/jsons_idx/TSPO_*_frameIdx.json): the formats of
tspo_amd.io / tspo_amd.video against REAL docs (tests/golden/published.json = per-file invariants + whole sample docs, written by
tests/golden/make_golden.py from the reference's files).  They cannot be reproduced offline (no weights / videos); what is pinned
is the join, the JSON text, the frame plan a consumer builds from them and which branch of the generator produced a short doc."""
import hashlib
import json
import math
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from tspo_amd import io as tio, video  # noqa: E402

REF = "/root/reference/evaluation"


@pytest.fixture(scope="module")
def pub():
    return json.load(open(os.path.join(HERE, "golden", "published.json")))


def _step(frames):
    s = 0
    for v in frames:
        s = math.gcd(s, int(v))
    return max(s, 1)


def test_file_invariants(pub):
    assert set(pub) == {"LongVideoBench", "MLVU", "VideoMME"}
    for ds, g in pub.items():
        assert tio.join_key(ds) == g["join_key"] and tio.DATASET_JSON[ds] == g["anno_file"]
        assert g["distinct_keys"] == g["n_docs"]                       # the join key identifies a doc
        assert g["all_ascending"] and g["all_distinct"] and g["all_float_integers"] and g["short_docs_are_all_candidates"]
        assert g["max_len"] == 64 and g["n_with_64"] + g["n_short"] == g["n_docs"]
    assert (pub["LongVideoBench"]["n_docs"], pub["LongVideoBench"]["n_with_64"], pub["LongVideoBench"]["n_short"]) == (1337, 976, 361)
    assert pub["MLVU"]["n_short"] == 0 and pub["VideoMME"]["n_short"] == 261


def test_join_reproduces_the_published_text_on_real_docs(pub, tmp_path):
    """change_score_tch.py:33-44 on the sample: annotation docs (= the published docs without their last field) + {key: frames}
    -> write_frame_idx_json -> the JSON text the reference wrote for those docs, byte for byte; docs without a result stay."""
    for ds, g in pub.items():
        docs = g["sample_docs"]
        assert all(list(d)[-1] == "frame_idx" for d in docs)
        anno = [{k: v for k, v in d.items() if k != "frame_idx"} for d in docs]
        results = {d[g["join_key"]]: d["frame_idx"] for d in docs}
        pk = tmp_path / f"x_{ds}_supp.pkl"
        tio.save_results_pickle(results, str(pk))
        out = tio.frame_idx_json_path(str(tmp_path / "evaluation"), "x", ds)
        n = tio.write_frame_idx_json(anno, tio.load_results_pickle(str(pk)), out, dataset=ds)
        assert n == len(docs) and open(out).read() == json.dumps(docs)
        held = dict(results)
        held.pop(docs[1][g["join_key"]])
        tio.write_frame_idx_json(anno, held, out, dataset=ds)
        back = json.load(open(out))
        assert "frame_idx" not in back[1] and back[0] == docs[0] and back[2:] == docs[2:]


def test_consumer_frame_plans_on_real_docs(pub):
    """llava_vid_tspo.py:362-380 / qwen25vl_vision_process.py:402-412 on real docs: 64 entries -> exactly those frames and times
    at the doc's candidate step; fewer than 64 -> the consumer falls back to a uniform plan (the reference's behaviour)."""
    for ds, g in pub.items():
        for d in g["sample_docs"]:
            f = video.doc_frame_idx(d)
            assert f == d["frame_idx"] and video.doc_frame_idx({"frame_idx": [f]}, batched=True) == f
            step = _step(f) if len(f) > 1 else 30
            total = int(f[-1]) + step                                   # a reader at least as long as the last selected frame
            plan = video.plan_from_index(total, float(step), 1, 64, f)
            if len(f) == 64:
                assert [float(i) for i in plan.frame_idx] == f
                assert plan.frame_time == ",".join(f"{v / step:.2f}s" for v in f)
            else:
                assert len(plan.frame_idx) == 64 and plan.frame_idx[0] == 0 and plan.frame_idx[-1] == total - 1
            assert plan.video_time == total / float(step)


def test_short_docs_come_from_the_no_selection_branch(pub):
    """gen_id_tspo.py:83-92: a video with no more than 64 one-fps candidates is not scored at all - `select_frame_ids` returns
    every candidate; that is exactly what every published doc with fewer than 64 entries holds."""
    class NoModel:
        def temporal_sampling(self, *a, **k):
            raise AssertionError("selection must not run for T <= sample_num")
    seen = 0
    for ds, g in pub.items():
        for d in g["sample_docs"]:
            f = d["frame_idx"]
            if len(f) >= 64:
                continue
            seen += 1
            T = len(f)
            sampled_idx = torch.tensor([int(v) for v in f])
            got = tio.select_frame_ids(NoModel(), torch.zeros(T, 8), torch.zeros(1, 8), torch.zeros(T), sampled_idx, ds)
            assert got == f and all(isinstance(v, float) for v in got)
    assert seen >= 6


def test_compare_tool_on_real_docs(pub, tmp_path):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import compare_frame_idx as cmp
    docs = pub["MLVU"]["sample_docs"]
    s = cmp.compare(docs, docs)
    assert s["join_key"] == "question_id" and s["exact_match_rate"] == 1.0 and s["mean_jaccard"] == 1.0 and s["docs_compared"] == len(docs)
    moved = json.loads(json.dumps(docs))
    st = _step(moved[0]["frame_idx"])
    moved[0]["frame_idx"][5] += st if moved[0]["frame_idx"][5] + st not in moved[0]["frame_idx"] else 7 * st   # one frame one step later
    moved[1]["frame_idx"] = moved[1]["frame_idx"][:32] + [v + 10000 * st for v in moved[1]["frame_idx"][32:]]    # half the frames elsewhere
    del moved[2]
    s = cmp.compare(moved, docs, worst=3)
    assert s["missing_in_produced"] == 1 and s["docs_compared"] == len(docs) - 1
    assert s["exact_match_rate"] == (len(docs) - 3) / (len(docs) - 1)
    w = {r["key"]: r for r in s["worst"]}
    assert abs(w[docs[1]["question_id"]]["jaccard"] - 32 / 96) < 1e-9 and w[docs[0]["question_id"]]["jaccard"] == 63 / 65
    pa, pb = tmp_path / "a.json", tmp_path / "b.json"
    pa.write_text(json.dumps(moved)); pb.write_text(json.dumps(docs))
    assert cmp.main([str(pa), str(pb)]) == 1 and cmp.main([str(pb), str(pb)]) == 0


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (authoring container only)")
def test_full_files_against_the_reference_checkout(pub, tmp_path):
    """Where the reference is present: the WHOLE published files are reproduced byte for byte by the join over the reference's own
    annotation files, and the committed fixture still describes them (sha256)."""
    for ds, g in pub.items():
        text = open(f"{REF}/jsons_idx/TSPO_{ds}_frameIdx.json").read()
        assert hashlib.sha256(text.encode()).hexdigest() == g["sha256"] and len(text) == g["bytes"]
        docs = json.loads(text)
        anno = json.load(open(f"{REF}/jsons/{g['anno_file']}.json"))
        out = tio.frame_idx_json_path(str(tmp_path), "TSPO", ds)
        n = tio.write_frame_idx_json(anno, {d[g["join_key"]]: d["frame_idx"] for d in docs}, out, dataset=ds)
        assert n == g["n_docs"] and open(out).read() == text
        assert [docs[i] for i in g["sample_positions"]] == g["sample_docs"]
