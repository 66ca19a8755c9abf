"""On-disk formats the reference's evaluation / checkpoint flow uses, so the native agent plugs into it:

* feature cache  - mp_tools/vlmeval/vlm/gen_id_tspo.py:68-79: torch.save({"image" [T,768], "text" [1,768],
                   "clip_scores" [T], "sampled_idx" [T] int64}, f"{save_root}/{dataset}/{index}.pth")
* frame indices  - gen_id_tspo.py:83-92: list of *float* absolute frame numbers; selection is skipped when
                   T <= sample_num; VideoMME uses 'bin-max', the other benchmarks 'topk'; window 12, 64 frames
* index JSON     - mp_tools/change_score_tch.py: each doc of evaluation/jsons/<ds>.json gets a "frame_idx" field
* checkpoints    - scripts/merge_weights.py:19-24: selector tensors live under the prefix "multiModal_align."
                   in training checkpoints and under "selector." in the merged TSPO-0.4B model
"""
from __future__ import annotations

import json
import os
import pickle
from typing import Callable, Dict, Iterable, List, Optional

import torch

SELECTOR_KEYS = (
    "temporal.Self_q.weight", "temporal.Self_q.bias", "temporal.Self_k.weight", "temporal.Self_k.bias",
    "temporal.Self_v.weight", "temporal.Self_v.bias", "temporal.ffn_o.weight", "temporal.ffn_o.bias",
    "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias",
)
TRAIN_PREFIX = "multiModal_align."
MERGED_PREFIX = "selector."
DATASET_METHOD = {"LongVideoBench": "topk", "MLVU": "topk", "LVBench": "topk", "VideoMME": "bin-max"}


def feature_cache_path(save_root: str, dataset: str, index) -> str:
    return os.path.join(save_root, dataset, f"{index}.pth")


def save_feature_cache(path: str, image, text, clip_scores, sampled_idx) -> None:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"image": image.cpu(), "text": text.cpu(), "clip_scores": clip_scores.cpu(),
                "sampled_idx": torch.as_tensor(sampled_idx)}, path)


def load_feature_cache(path: str, device=None, dtype=None):
    """gen_id_tspo.py:74-79: tensors go `.to(device).to(dtype)` of the model; sampled_idx stays on the host."""
    stat = torch.load(path, map_location="cpu")
    img, txt, clip, idx = stat["image"], stat["text"], stat["clip_scores"], stat["sampled_idx"]
    if device is not None:
        img, txt, clip = (t.to(device) if dtype is None else t.to(device).to(dtype) for t in (img, txt, clip))
    return img, txt, clip, idx


def select_frame_ids(model, image, text, clip_scores, sampled_idx, dataset: str, sample_num: int = 64,
                     window_size: int = 12) -> List[float]:
    """gen_id_tspo.py:81-92 with `model` = tspo_amd.temporal_agent.TSPOModel (or anything exposing
    temporal_sampling)."""
    assert dataset in DATASET_METHOD
    if len(image) > sample_num:
        with torch.no_grad():
            ts_ids, _ = model.temporal_sampling(image, text, clip_scores, DATASET_METHOD[dataset], window_size, sample_num)
        abs_ids = torch.as_tensor(sampled_idx)[ts_ids.cpu()]
    else:
        abs_ids = torch.as_tensor(sampled_idx)
    return abs_ids.float().tolist()


# mp_tools/change_score_tch.py:22-38: annotation file and the field the per-sample results are joined on
DATASET_JSON = {"VideoMME": "videomme", "LongVideoBench": "lvb_val", "MLVU": "mlvu"}
DATASET_JOIN_KEY = {"VideoMME": "question_id", "MLVU": "question_id", "LongVideoBench": "id"}


def join_key(dataset: str) -> str:
    """Field of an annotation entry that identifies a sample of `dataset` (change_score_tch.py:33-38)."""
    for name, key in DATASET_JOIN_KEY.items():
        if name.lower() == dataset.lower():
            return key
    raise NotImplementedError(f"join key of dataset {dataset!r}: to be implemented")   # the reference raises the same


def load_results_pickle(path: str) -> Dict:
    """`work_dir/{name}_{data}_supp.pkl` of the harness: {index: [float absolute frame numbers]} (change_score_tch.py:30-31)."""
    with open(path, "rb") as f:
        return pickle.load(f)


def save_results_pickle(results: Dict, path: str) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump({k: [float(x) for x in v] for k, v in results.items()}, f)


def write_frame_idx_json(docs: Iterable[dict], results: Dict, out_path: str, key: Optional[str] = None,
                         dataset: Optional[str] = None) -> int:
    """change_score_tch.py:33-44: adds "frame_idx" to every annotation entry whose join key (by `dataset`, or the explicit
    `key`) is in `results` ({index: [float,...]}, e.g. from load_results_pickle) and writes the JSON list; entries
    without a result are kept unchanged (the reference prints their index and continues).  Returns the number joined."""
    if key is None:
        key = join_key(dataset) if dataset is not None else "index"
    out, n = [], 0
    for d in docs:
        d = dict(d)
        if d.get(key) in results:
            d["frame_idx"] = [float(x) for x in results[d[key]]]
            n += 1
        out.append(d)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(out, f)
    return n


def frame_idx_json_path(base_anno_path: str, name: str, dataset: str) -> str:
    """change_score_tch.py:27: evaluation/jsons_idx/{name}_{data}_frameIdx.json"""
    return os.path.join(base_anno_path, "jsons_idx", f"{name}_{dataset}_frameIdx.json")


class FrameIdGenerator:
    """`GEN_Frame_ID_TSPO.generate_inner` (mp_tools/vlmeval/vlm/gen_id_tspo.py:59-92) around a TSPOModel:
    cache miss -> decode the video (`load_video`, a plug-in: decord is outside this package), `extract_feature`, save the
    .pth cache; cache hit -> load it, move to the model's device and dtype; then `temporal_sampling` (bin-max for
    VideoMME, top-k otherwise, window 12) unless the video has no more than `sample_num` frames; returns the selected
    absolute frame numbers as floats."""

    def __init__(self, model, clip_processor, save_root: str, sample_num: int = 64,
                 load_video: Optional[Callable] = None):
        self.tspo_model, self.clip_processor = model, clip_processor
        self.save_root, self.sample_num, self.load_video = save_root, sample_num, load_video
        self.cache_hits = self.cache_misses = 0

    @staticmethod
    def problem_of(question: str) -> str:
        assert "\nOptions" in question
        return question.replace("<image>\n", "").replace("Question: ", "").split("\nOptions")[0]

    def generate_inner(self, message, index=None, dataset=None) -> List[float]:
        import torch
        video_path, question = message[0]["value"], message[1]["value"]
        problem = self.problem_of(question)
        path = feature_cache_path(self.save_root, dataset, index)
        model = self.tspo_model
        if not os.path.exists(path):
            if self.load_video is None:
                raise FileNotFoundError(f"{path} is not cached and no load_video plug-in was given")
            video, _, _, sampled_idx = self.load_video(video_path, max_frames_num=50000, fps=1, force_sample=False)
            image, text, clip = model.extract_feature(self.clip_processor, video, problem)
            save_feature_cache(path, image, text, clip, sampled_idx)
            sampled_idx = torch.as_tensor(sampled_idx)
            self.cache_misses += 1
        else:
            image, text, clip, sampled_idx = load_feature_cache(path, model.device, model.dtype)
            self.cache_hits += 1
        return select_frame_ids(model, image, text, clip, sampled_idx, dataset, self.sample_num)


def extract_selector_state(state: Dict[str, torch.Tensor], prefix: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """Pull the 12 selector tensors out of a (sharded-checkpoint) state dict (merge_weights.py:19-24)."""
    prefixes = [prefix] if prefix is not None else [TRAIN_PREFIX, MERGED_PREFIX, "model." + TRAIN_PREFIX, ""]
    for p in prefixes:
        sub = {k[len(p):]: v for k, v in state.items() if k.startswith(p) and k[len(p):] in SELECTOR_KEYS}
        if len(sub) == len(SELECTOR_KEYS):
            return sub
    raise KeyError("no complete selector state (12 tensors) found under prefixes " + repr(prefixes))


def save_selector_safetensors(selector_state: Dict[str, torch.Tensor], path: str, prefix: str = TRAIN_PREFIX) -> None:
    from safetensors.torch import save_file
    save_file({prefix + k: v.detach().cpu().contiguous() for k, v in selector_state.items()}, path)


def load_selector_safetensors(path: str) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        state = {k: f.get_tensor(k) for k in f.keys()}
    return extract_selector_state(state)
