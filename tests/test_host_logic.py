"""Host-side logic of the drop-in API (no GPU): reference signatures, state-dict keys, flat bucket,
rewards, file formats, AKS / anchors against the golden vectors."""
import inspect
import json
import os

import numpy as np
import pytest
import torch

from tspo_amd import io as tio, ops, rewards, synth
from tspo_amd.pipeline import annealed_tau
from tspo_amd.temporal_agent import MultiModal_Align, TSPOModel, positional_encoding
from tspo_amd.utils import AKS_sampling, generate_uniform_integers, gumbel_softmax
from tspo_amd.policy import TemporalPolicy


def test_signatures_match_reference():
    """the Python surface to reproduce (names, argument order, defaults)."""
    def params(f):
        return [(p.name, p.default) for p in inspect.signature(f).parameters.values()]
    E = inspect.Parameter.empty
    assert params(positional_encoding) == [("T", E), ("C", E)]
    assert params(MultiModal_Align.__init__)[1:] == [("dim", 768), ("num_heads", 8), ("dropout", 0.0), ("gamma", 0.6), ("bias", 0.2)]
    assert params(MultiModal_Align.forward)[1:] == [("input_emb", E), ("text_emb", E), ("clip_scores", None),
                                                    ("window_size", None), ("score_tau", 0.025)]
    assert params(MultiModal_Align.create_window_mask)[1:] == [("seq_len", E), ("window_size", 8)]
    assert params(TSPOModel.extract_feature)[1:] == [("clip_processor", E), ("candidates", E), ("problem", E), ("processor_type", "llava")]
    assert params(TSPOModel.temporal_sampling)[1:] == [("image_features", E), ("text_features", E), ("clip_scores", E),
                                                       ("method", E), ("window_size", E), ("sample_num", E)]
    assert params(TSPOModel.forward)[1:] == [("clip_processor", E), ("candidates", E), ("problem", E), ("sample_num", E),
                                             ("window_size", 12), ("method", "topk"), ("processor_type", "llava")]
    assert params(TSPOModel.inference_ts)[1:] == [("confidence", E), ("method", E), ("sample_len", E)]
    assert params(gumbel_softmax)[:3] == [("logits", E), ("tau", 1.0), ("sample_len", 64)]
    assert params(TemporalPolicy.temporal_sampling)[1:9] == [
        ("image_embeddings", E), ("text_features", E), ("clip_scores", None), ("sample_len", 64), ("ts_ids", None),
        ("window_size", None), ("score_tau", 0.025), ("method", None)]
    assert hasattr(TSPOModel, "from_merged_components") and hasattr(TSPOModel, "save_pretrained")


def test_selector_state_dict_keys_and_flat_bucket():
    m = MultiModal_Align()
    keys = sorted(m.state_dict().keys())
    assert keys == sorted(tio.SELECTOR_KEYS)
    assert sum(p.numel() for p in m.parameters()) == 3_543_552
    assert ops.trainable_numel(768) == 2_952_960
    st = {k: torch.from_numpy(v) for k, v in synth.selector_state(64, seed=3, bias_std=0.1).items()}
    m = MultiModal_Align(dim=64)
    m.load_state_dict(st)
    flat = m.flatten_parameters("cpu")
    offs = ops.flat_offsets(64)
    for k, v in st.items():
        off, shape = offs[k]
        assert torch.equal(flat[off:off + v.numel()].view(shape), v)
        assert dict(m.named_parameters())[k].data_ptr() == flat.data_ptr() + 4 * off       # views, not copies
    assert m._is_flat()
    # q|k|v weights contiguous -> fused [3D, D] projection
    assert offs["temporal.Self_k.weight"][0] == 64 * 64 and offs["temporal.Self_v.weight"][0] == 2 * 64 * 64
    m.to(torch.bfloat16)
    assert not m._is_flat()                                                      # falls back to a packed fp32 copy
    assert torch.allclose(m._flat_params()[:64 * 64].view(64, 64), st["temporal.Self_q.weight"], atol=1e-2)


def test_reference_error_behaviour_without_gpu():
    m = MultiModal_Align(dim=64)
    x, t, c = torch.zeros(5, 64), torch.zeros(1, 64), torch.zeros(5)
    with pytest.raises(TypeError):
        m(x, t, c, window_size=None)              # reference: None // 2
    with pytest.raises(TypeError):
        m(x, t, None, window_size=12)             # reference: tensor + None
    with pytest.raises(RuntimeError):
        gumbel_softmax(torch.zeros(4, 1), sample_len=5)    # torch.topk: k out of range
    with pytest.raises(UnboundLocalError):
        TSPOModel.inference_ts(None, torch.zeros(4), "nope", 2)


def test_window_mask_and_posenc_match_golden(golden):
    g = golden["selector"]
    m = MultiModal_Align(dim=64)
    from inputs import SELECTOR_CASES
    for name, T, D, H, w, tau, M, ks in SELECTOR_CASES:
        if f"{name}.mask" in g.files:
            np.testing.assert_array_equal(m.create_window_mask(T, w).numpy().astype(np.uint8), g[f"{name}.mask"])
    np.testing.assert_array_equal(positional_encoding(4, 8).numpy(), golden["misc"]["pe_4_8"])


def test_aks_and_anchors_match_golden(golden, capsys):
    g = golden["selector"]
    for key in golden["misc"].files:
        if key.startswith("uni_"):
            _, t, l = key.split("_")
            assert generate_uniform_integers(int(t), int(l)) == golden["misc"][key].tolist()
    for name in ("s32", "s50", "s1024"):
        s = g[f"{name}.scores"]
        for k in (8, 16):
            assert AKS_sampling(s.astype(np.float32), k) == g[f"{name}.aks{k}"].tolist()


def test_rewards():
    comp = [[{"content": "The answer is (B)."}], [{"content": "c"}], [{"content": "no option"}]]
    sol = ["<answer>B</answer>", "<answer>B</answer>", "A"]
    assert rewards.accuracy_reward(comp, sol) == [1.0, 0.0, 0.0]
    assert rewards.map_prediction_to_option("Answer: D") == "d" and rewards.map_prediction_to_option("xyz") is False
    mask = torch.tensor([True, False, True, True, False, False])
    ids = [(None, torch.tensor([0, 1, 2])), (None, torch.tensor([3, 4, 5]))]
    assert rewards.temporal_localization_reward(None, None, ids, mask) == [2 / 3, 1 / 3]
    assert rewards.format_reward([[{"content": "<think>a</think> <answer>b</answer>"}], [{"content": "b"}]]) == [1.0, 0.0]
    rpf = torch.tensor([[1.0, 0.5], [0.0, 0.25]])
    assert rewards.combine_rewards(rpf, "specific").tolist() == [1.5, 0.25]
    assert rewards.combine_rewards(rpf, "general").tolist() == [2.0, 1.0]
    assert rewards.training_sample_len(16, "general") == 8 and rewards.training_sample_len(16, "specific") == 16
    idx = torch.tensor([[[0, 2, 3], [1, 4, 5]]])
    np.testing.assert_allclose(rewards.selection_mask_reward_gpu(idx, mask[None]).numpy(), [[1.0, 0.0]])
    assert abs(annealed_tau(0.025, 50, 100) - 0.0175) < 1e-12


def test_io_formats(tmp_path):
    img, txt, clip = torch.randn(70, 768), torch.randn(1, 768), torch.randn(70)
    sidx = torch.arange(0, 700, 10)
    p = tio.feature_cache_path(str(tmp_path), "MLVU", 3)
    tio.save_feature_cache(p, img, txt, clip, sidx)
    stat = torch.load(p)
    assert sorted(stat) == ["clip_scores", "image", "sampled_idx", "text"]        # the reference's keys
    i2, t2, c2, s2 = tio.load_feature_cache(p)
    assert torch.equal(i2, img) and torch.equal(s2, sidx)

    class Fake:
        def temporal_sampling(self, image, text, clip_scores, method, window_size, sample_num):
            assert method == "bin-max" and window_size == 12 and sample_num == 64
            return torch.arange(64), clip_scores
    ids = tio.select_frame_ids(Fake(), img, txt, clip, sidx, "VideoMME")
    assert ids == [float(10 * i) for i in range(64)] and all(isinstance(x, float) for x in ids)
    assert tio.select_frame_ids(Fake(), img[:10], txt, clip[:10], sidx[:10], "MLVU") == sidx[:10].float().tolist()
    out = tmp_path / "idx.json"
    n = tio.write_frame_idx_json([{"index": 1, "q": "a"}, {"index": 2}], {1: ids}, str(out))
    docs = json.load(open(out))
    assert n == 1 and docs[0]["frame_idx"] == ids and "frame_idx" not in docs[1]

    st = {k: torch.from_numpy(v) for k, v in synth.selector_state(64, seed=9).items()}
    ck = tmp_path / "model-00004-of-00004.safetensors"
    tio.save_selector_safetensors(st, str(ck), prefix=tio.TRAIN_PREFIX)
    back = tio.load_selector_safetensors(str(ck))
    assert sorted(back) == sorted(st) and all(torch.equal(back[k], st[k]) for k in st)
    merged = {tio.MERGED_PREFIX + k: v for k, v in st.items()}
    merged["vision_model.x"] = torch.zeros(1)
    assert sorted(tio.extract_selector_state(merged)) == sorted(st)
    m = MultiModal_Align(dim=64)
    m.load_state_dict(back)


def test_reference_import_aliases():
    import tspo_amd
    tspo_amd.install_reference_aliases()
    from model.temporal_agent import TSPOModel as T2, MultiModal_Align as M2    # the reference's import path
    from model.utils import gumbel_softmax as g2
    assert T2 is TSPOModel and M2 is MultiModal_Align and g2 is gumbel_softmax


def _tiny_clip_config():
    from transformers import CLIPConfig
    return CLIPConfig(text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                                       vocab_size=49408, max_position_embeddings=77, projection_dim=768),
                      vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                         image_size=224, patch_size=14, projection_dim=768), projection_dim=768)


def test_hf_entry_points_and_merge_flow(tmp_path):
    """The calls the reference's callers make: gen_id_tspo.py:55 (`from_pretrained(path, attn_implementation=
    "flash_attention_2", torch_dtype=torch.bfloat16, device_map="auto")` - flash_attn is not installed in the ROCm image)
    and scripts/merge_weights.py:14-58 (selector tensors out of the last of four training shards ->
    from_merged_components -> save_pretrained -> reload)."""
    from safetensors.torch import save_file
    torch.manual_seed(0)
    model = TSPOModel(_tiny_clip_config())
    sel = {k: torch.from_numpy(v) for k, v in synth.selector_state(768, seed=5, std=0.02, bias_std=0.01).items()}
    model.selector.load_state_dict(sel)
    d1 = tmp_path / "tspo"
    model.save_pretrained(str(d1))
    back = TSPOModel.from_pretrained(str(d1), attn_implementation="flash_attention_2", torch_dtype=torch.bfloat16,
                                     device_map="auto")
    assert isinstance(back, TSPOModel) and back.dtype == torch.bfloat16 and isinstance(back.selector, MultiModal_Align)
    sd1, sd2 = model.state_dict(), back.state_dict()
    assert sorted(sd1) == sorted(sd2) and len([k for k in sd2 if k.startswith("selector.")]) == 12
    assert all(torch.equal(sd1[k].to(torch.bfloat16), sd2[k]) for k in sd1)

    # a DeepSpeed-style training checkpoint: four shards, the selector ("multiModal_align.*") in the last one
    shards = tmp_path / "train_ckpt"
    shards.mkdir()
    for i in range(1, 5):
        st = {f"model.layers.{i}.dummy.weight": torch.randn(4, 4)}
        if i == 4:
            st.update({tio.TRAIN_PREFIX + k: v.clone() for k, v in sel.items()})
        save_file(st, str(shards / f"model-0000{i}-of-00004.safetensors"))
    from transformers import CLIPModel
    clip_model = CLIPModel(_tiny_clip_config()).to(torch.bfloat16)
    module_state_dict = tio.load_selector_safetensors(str(shards / "model-00004-of-00004.safetensors"))
    assert sorted(module_state_dict) == sorted(sel)
    merged = TSPOModel.from_merged_components(clip_model=clip_model, selector_state_dict=module_state_dict)
    assert merged.dtype == torch.bfloat16
    d2 = tmp_path / "TSPO-0.4B"
    merged.save_pretrained(str(d2))
    again = TSPOModel.from_pretrained(str(d2), attn_implementation="flash_attention_2", torch_dtype=torch.bfloat16,
                                      device_map="auto")
    for k, v in sel.items():
        assert torch.equal(again.selector.state_dict()[k], v.to(torch.bfloat16))
    for k, v in clip_model.state_dict().items():
        assert torch.equal(again.state_dict()[k], v)
    assert sorted(tio.extract_selector_state(again.state_dict())) == sorted(sel)      # "selector." prefix of the merged model


def test_frame_idx_join_and_pickle(tmp_path):
    """mp_tools/change_score_tch.py:22-44: results pickle {index: [float]} joined into the annotation list on
    `question_id` (VideoMME, MLVU) / `id` (LongVideoBench); unknown datasets raise like the reference."""
    assert tio.join_key("VideoMME") == "question_id" and tio.join_key("mlvu") == "question_id"
    assert tio.join_key("LongVideoBench") == "id"
    with pytest.raises(NotImplementedError):
        tio.join_key("LVBench")
    res = {"q1": [0.0, 30.0, 60.5], 17: [5, 9]}
    pk = tmp_path / "work_dir" / "run_VideoMME_supp.pkl"
    tio.save_results_pickle(res, str(pk))
    assert tio.load_results_pickle(str(pk)) == {"q1": [0.0, 30.0, 60.5], 17: [5.0, 9.0]}
    anno = [{"question_id": "q1", "video": "a.mp4"}, {"question_id": "q2", "video": "b.mp4"}]
    out = tio.frame_idx_json_path(str(tmp_path), "run", "VideoMME")
    assert out.endswith(os.path.join("jsons_idx", "run_VideoMME_frameIdx.json"))
    assert tio.write_frame_idx_json(anno, tio.load_results_pickle(str(pk)), out, dataset="VideoMME") == 1
    docs = json.load(open(out))
    assert docs[0]["frame_idx"] == [0.0, 30.0, 60.5] and "frame_idx" not in docs[1] and docs[1]["video"] == "b.mp4"
    anno2 = [{"id": 17}, {"id": 18}]
    assert tio.write_frame_idx_json(anno2, tio.load_results_pickle(str(pk)), str(tmp_path / "l.json"), dataset="LongVideoBench") == 1
    assert tio.FrameIdGenerator.problem_of("<image>\nQuestion: what happens?\nOptions:\nA. x") == "what happens?"


def test_bench_accounting_helpers():
    """The FLOP / byte accounting bench.py divides by: 162.03 GFLOP per CLIP-L/14 frame, of which 155.53 are
    GEMM FLOPs; algorithmic GEMM bytes per launch; the policy-step roofline block."""
    import bench
    c = bench.CLIP_L14
    g, a = bench.gemm_flops_per_frame(c), bench.attn_flops_per_frame(c)
    assert abs(g / 1e9 - 155.53) < 0.01 and abs(a / 1e9 - 6.49) < 0.01 and abs((g + a) / 1e9 - 162.03) < 0.02
    assert abs(bench.alg_bytes_per_launch(c, 1024) / 1e9 - 2.39) < 0.01
    r = bench.policy_step_roofline(4, 512, 768, 4e-4)
    assert r["bound"] == "mfma" and r["dtype"] == "f32" and r["launches_per_step"] is None      # counted live on the GPU only
    assert bench.policy_step_roofline(4, 512, 768, 4e-4, 15, {"x": 15.0})["launches_per_step"] == 15
    assert abs(r["gemm_flop_per_step"] / 1e9 - 28.99) < 0.01
    assert abs(r["achieved"] - 28.99e9 / 4e-4 / 1e12) < 0.1 and abs(r["frac"] - r["achieved"] / 157.3) < 1e-3


def test_reproduce_published_idx_tool_host_logic():
    """tools/reproduce_published_idx.py (local weights + videos -> FrameIdGenerator -> compare against the published lists): its
    GPU / decord branch cannot run here; the doc -> (join key, video file, problem text) mapping can, on docs shaped like the
    reference's three annotation files, and the problem text must survive FrameIdGenerator.problem_of (gen_id_tspo.py:63-64)."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("reproduce_published_idx", os.path.join(root, "tools", "reproduce_published_idx.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from tspo_amd.io import FrameIdGenerator, join_key
    docs = {"LongVideoBench": {"id": "86C_0", "video_path": "86C.mp4", "question": "Which subtitles appear at the same time?"},
            "MLVU": {"question_id": "Q0", "video_name": "needle_32.mp4", "question": "What does the hand do?\n(A) Delivers\n(B) Shakes"},
            "VideoMME": {"question_id": "001-1", "videoID": "fFjv93ACGo8", "question": "When is the tree decorated?"}}
    want = {"LongVideoBench": ("86C.mp4", "Which subtitles appear at the same time?"), "MLVU": ("needle_32.mp4", "What does the hand do?"),
            "VideoMME": ("fFjv93ACGo8.mp4", "When is the tree decorated?")}
    for ds, d in docs.items():
        fname, key, video_of = m.ANNO[ds]
        assert key == join_key(ds) and key in d
        assert (video_of(d), m.problem_of(d)) == want[ds]
        assert FrameIdGenerator.problem_of("Question: " + m.problem_of(d) + "\nOptions") == want[ds][1]


def test_batch_meta_is_per_prompt_stacked_or_not():
    """a reward plug-in reading batch.meta must see ONE convention - a list with one entry per prompt - on a single
    micro-batch and on the stacked batch of a coalesced accumulation window (tspo_amd.train._stack)."""
    import pytest as _pt
    from tspo_amd import train as TR

    def mk(B, meta):
        return TR.Batch(torch.zeros(B, 6, 4), torch.zeros(B, 1, 4), torch.zeros(B, 6), torch.ones(B, 6, dtype=torch.bool), "specific", meta)
    a, b = mk(1, {"path": "a.pth"}), mk(1, [{"path": "b.pth"}])
    assert a.meta == [{"path": "a.pth"}] and b.meta == [{"path": "b.pth"}] and mk(2, None).meta == [None, None]
    st = TR._stack([a, b])
    assert st.feats.shape[0] == 2 and st.meta == [{"path": "a.pth"}, {"path": "b.pth"}] and TR._stackable([a, b])
    st2 = TR._stack([mk(2, [1, 2]), mk(2, [3, 4])])
    assert st2.meta == [1, 2, 3, 4] and st2.feats.shape[0] == 4
    with _pt.raises(ValueError):
        mk(2, [1])
    with _pt.raises(ValueError):
        mk(2, {"path": "x"})
    seen = []

    def reward(idx, batch):           # a plug-in that indexes meta per prompt, as a video-LLM reward pass would (its video path)
        seen.append([batch.meta[i] for i in range(idx.shape[0])])
        return torch.zeros(idx.shape[0], idx.shape[1], 2)
    reward(torch.zeros(2, 3, 2, dtype=torch.long), st)
    reward(torch.zeros(1, 3, 2, dtype=torch.long), a)
    assert seen == [[{"path": "a.pth"}, {"path": "b.pth"}], [{"path": "a.pth"}]]
