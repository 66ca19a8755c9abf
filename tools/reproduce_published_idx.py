#!/usr/bin/env python
"""One command from LOCAL weights + videos to the acceptance check against the reference's PUBLISHED frame lists
(evaluation/jsons_idx/TSPO_<dataset>_frameIdx.json) - for whoever has what this repository cannot ship: the TSPO-0.4B
checkpoint (CLIP-L/14 + `selector.*`, scripts/merge_weights.py of the reference), the benchmark videos and decord.

    python tools/reproduce_published_idx.py --dataset MLVU --weights /data/TSPO-0.4B --videos /data/MLVU/video \
        --reference /path/to/TSPO [--limit 50] [--save-root work_dir/feats] [--out work_dir/ours_MLVU_frameIdx.json]

What runs, in the reference's order (mp_tools/vlmeval/vlm/gen_id_tspo.py:51-92, mp_tools/change_score_tch.py:22-44):
  TSPOModel.from_pretrained(weights, bf16) + CLIPProcessor.from_pretrained(weights)              gen_id_tspo.py:55-56
  per annotation doc of <reference>/evaluation/jsons/{lvb_val,mlvu,videomme}.json:
    1-fps candidates of the video (decord; tspo_amd.video.plan_uniform, max 50000)               gen_id_tspo.py:70
    tspo_amd.io.FrameIdGenerator.generate_inner: HIP preprocessing + CLIP encode + scoring head + top-k / bin-max,
    the .pth feature cache in the reference's format                                              gen_id_tspo.py:68-92
  tspo_amd.io.write_frame_idx_json -> --out (byte-compatible with the published files)            change_score_tch.py
  tools/compare_frame_idx.py --out vs <reference>/evaluation/jsons_idx/TSPO_<dataset>_frameIdx.json
Exit code 0 and one JSON summary line (exact-match docs, mean Jaccard, within-one-step overlap).

The `problem` string the agent sees is the annotation's question up to its options, which is what gen_id_tspo.py:64 cuts out of
the harness's prompt ("Question: ...\nOptions ...").  What IS verified here: `run()` end to end on the GPU with a small random
TSPOModel checkpoint written by save_pretrained, a stub processor and an in-memory video reader
(tests/test_gpu_flow.py::test_reproduce_published_idx_flow_round_trip: produce -> publish -> reproduce = exact match, second pass
from the feature caches).  What is NOT: the real checkpoint, real videos and decord - none of them is in this image."""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ANNO = {"LongVideoBench": ("lvb_val.json", "id", lambda d: d["video_path"]),
        "MLVU": ("mlvu.json", "question_id", lambda d: d["video_name"]),
        "VideoMME": ("videomme.json", "question_id", lambda d: d["videoID"] + ".mp4")}


def problem_of(doc: dict) -> str:
    """The text the selector is conditioned on: the question without its options (gen_id_tspo.py:64)."""
    q = doc["question"]
    for cut in ("\nOptions", "\n(A)", "\nA."):
        if cut in q:
            q = q.split(cut)[0]
    return q.replace("<image>\n", "").replace("Question: ", "").strip()


def run(dataset, weights, videos, reference, save_root, out=None, limit=0, sample_num=64, processor=None, open_video=None,
        published_name=None):
    """The whole flow; returns compare_frame_idx's summary.  `processor` (default CLIPProcessor.from_pretrained(weights)) and
    `open_video(path) -> reader with len() / get_avg_fps() / get_batch(idx).asnumpy()` (default decord.VideoReader) are
    injectable, which is how tests/test_gpu_flow.py runs this flow end to end without the checkpoint's tokenizer files or decord."""
    import torch
    from tspo_amd import io as tio
    from tspo_amd import video as tvideo
    from tspo_amd.temporal_agent import TSPOModel
    import compare_frame_idx as cmp

    if open_video is None:
        try:
            from decord import VideoReader, cpu
        except ImportError:
            sys.exit("decord is not installed (the reference's video reader, requirements.txt); install it next to the videos")
        open_video = lambda path: VideoReader(path, ctx=cpu(0), num_threads=1)      # noqa: E731
    if processor is None:
        from transformers import CLIPProcessor
        processor = CLIPProcessor.from_pretrained(weights)
    fname, key, video_of = ANNO[dataset]
    docs = json.load(open(os.path.join(reference, "evaluation", "jsons", fname)))
    if limit:
        docs = docs[:limit]
    published = json.load(open(os.path.join(reference, "evaluation", "jsons_idx", published_name or f"TSPO_{dataset}_frameIdx.json")))
    model = TSPOModel.from_pretrained(weights, torch_dtype=torch.bfloat16).to("cuda").eval()      # gen_id_tspo.py:55

    def load_video(path, max_frames_num=50000, fps=1, force_sample=False):
        vr = open_video(path)
        plan = tvideo.plan_uniform(len(vr), vr.get_avg_fps(), fps, max_frames_num, force_sample)
        return tvideo.load_frames(vr, plan), plan.frame_time, plan.video_time, torch.tensor(plan.frame_idx)

    gen = tio.FrameIdGenerator(model, processor, save_root, sample_num=sample_num, load_video=load_video)
    results = {}
    for i, d in enumerate(docs):
        msg = [{"type": "video", "value": os.path.join(videos, video_of(d))},
               {"type": "text", "value": "Question: " + problem_of(d) + "\nOptions"}]
        results[d[key]] = gen.generate_inner(msg, index=d[key], dataset=dataset)
        if (i + 1) % 25 == 0:
            print(f"  {i + 1}/{len(docs)} docs ({gen.cache_hits} cache hits)", file=sys.stderr, flush=True)
    out = out or os.path.join("work_dir", f"tspo_amd_{dataset}_frameIdx.json")
    tio.write_frame_idx_json(docs, results, out, key=key)
    pub = [d for d in published if d[key] in results]
    summary = cmp.compare(json.load(open(out)), pub, key=key)
    summary.update(out=out, cache_hits=gen.cache_hits, cache_misses=gen.cache_misses)
    return summary


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--dataset", required=True, choices=sorted(ANNO))
    ap.add_argument("--weights", required=True, help="TSPO-0.4B directory (HF format: CLIP-L/14 + selector.*)")
    ap.add_argument("--videos", required=True, help="directory holding the dataset's video files")
    ap.add_argument("--reference", required=True, help="checkout of Hui-design/TSPO (for evaluation/jsons and jsons_idx)")
    ap.add_argument("--save-root", default="work_dir/tspo_amd_feats", help=".pth feature caches (the reference's format)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--limit", type=int, default=0, help="first N docs only (0 = all)")
    ap.add_argument("--sample-num", type=int, default=64)
    a = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        sys.exit("needs an MI355X: the product has no CPU path")
    summary = run(a.dataset, a.weights, a.videos, a.reference, a.save_root, a.out, a.limit, a.sample_num)
    print(json.dumps(summary))
    return 1 if (summary["missing_in_produced"] or summary["extra_in_produced"]) else 0


if __name__ == "__main__":
    sys.exit(main())
