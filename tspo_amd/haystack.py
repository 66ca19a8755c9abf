"""Needle-in-a-haystack sample builder of the TSPO training step, at tensor level.

Reference: src/open_tspo/trainer/utils.py:15-25 (repeat_videos), :177-200 (shuffle_clips) and their call site
src/open_tspo/trainer/tspo_trainer.py:462-482: for a "specific" item the video the policy sees is 1-4 sub-sampled copies
("true clips", 50 frames each) of the question's video mixed with 12 clips of unrelated videos in random order;
`shuffle_mask[t]` says whether frame t comes from the question's video and feeds `temporal_localization_reward`
(src/open_tspo/tspo.py:146-159).

Same function names, argument meaning and RANDOM STREAM as the reference: the reference draws from numpy's global
legacy generator (np.random.choice / np.random.permutation); here the generator is an explicit
`np.random.RandomState` (`rng`), so `RandomState(seed)` reproduces the reference under `np.random.seed(seed)` draw for
draw (pinned by tests/golden/glue.json, generated from the reference's own functions).  Frames may be numpy arrays
(as in the reference) or torch tensors on any device - the mixing itself is an index gather, so on the GPU the frames
never leave HBM.
"""
from __future__ import annotations

import random as _random
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def _rng(rng) -> np.random.RandomState:
    if rng is None:
        return np.random.mtrand._rand          # numpy's global legacy generator, exactly what the reference uses
    if isinstance(rng, (int, np.integer)):
        return np.random.RandomState(int(rng))
    return rng


def _take(video, indices):
    if isinstance(video, torch.Tensor):
        return video[torch.as_tensor(indices, dtype=torch.long, device=video.device)]
    return video[indices]


def repeat_videos(video, repeat_times: int = 4, sample_len: int = 50, rng=None) -> List:
    """`repeat_times` sorted random sub-samples of `sample_len` frames (the whole video when it is not longer).
    video [L, h, w, 3] (or any [L, ...])."""
    r = _rng(rng)
    if video.shape[0] <= sample_len:
        return [video for _ in range(repeat_times)]
    samples = []
    for _ in range(repeat_times):
        indices = np.sort(r.choice(video.shape[0], size=sample_len, replace=False))
        samples.append(_take(video, indices))
    return samples


def shuffle_clips(true_groups: Sequence, wrong_groups: Sequence, t: int = 1, rng=None):
    """Random interleaving of the true and the distractor clips (all of one length).
    Returns (merged_video [G*len, ...], shuffle_mask bool tensor [G*len]: True = frame of a true clip)."""
    r = _rng(rng)
    len_group = len(true_groups[0])
    total_groups = [1] * len(true_groups) + [0] * len(wrong_groups)
    order = r.permutation(total_groups)
    parts, mask = [], np.zeros(len(total_groups) * len_group, dtype=bool)
    count_ori = count_wrong = 0
    for i, flag in enumerate(order):
        if flag == 1:
            parts.append(true_groups[count_ori])
            mask[i * len_group:(i + 1) * len_group] = True
            count_ori += 1
        else:
            parts.append(wrong_groups[count_wrong])
            count_wrong += 1
    if isinstance(parts[0], torch.Tensor):
        dev = parts[0].device
        merged = torch.cat([p.to(dev) for p in parts], dim=0)
    else:
        merged = np.concatenate(parts)
    return merged, torch.tensor(mask)


def build_specific_sample(video, distractor_clips: Sequence, repeat_times: Optional[int] = None, sample_len: int = 50,
                          rng=None, py_random: Optional[_random.Random] = None):
    """tspo_trainer.py:467-481 for a "specific" item: `repeat_times` true clips (default: random.randint(1, 4) like the
    reference) + the given distractor clips (the reference samples 12 of them from unrelated videos, each as long as a true
    clip), shuffled.  Returns (video [T, ...], shuffle_mask [T] bool)."""
    if repeat_times is None:
        repeat_times = (py_random or _random).randint(1, 4)
    true_videos = repeat_videos(video, repeat_times=repeat_times, sample_len=sample_len, rng=rng)
    glen = len(true_videos[0])
    for c in distractor_clips:
        if len(c) != glen:
            raise ValueError(f"distractor clip of {len(c)} frames, true clips have {glen}")
    return shuffle_clips(true_videos, list(distractor_clips), rng=rng)


def general_sample_mask(video) -> torch.Tensor:
    """tspo_trainer.py:482: "general" items keep the video as it is; every frame counts as relevant."""
    return torch.ones(len(video)).bool()


def synthetic_distractors(n_clips: int, clip_len: int, like, seed: int = 0) -> List:
    """Stand-in for `sample_real_frames` (trainer/utils.py, needs decord + a video corpus): random uint8 clips of the
    frame geometry of `like` [L, ...] on its device."""
    out = []
    for i in range(n_clips):
        if isinstance(like, torch.Tensor):
            g = torch.Generator(device=like.device).manual_seed(seed * 1000 + i)
            out.append(torch.randint(0, 256, (clip_len,) + tuple(like.shape[1:]), generator=g, device=like.device,
                                     dtype=torch.uint8).to(like.dtype))
        else:
            r = np.random.RandomState(seed * 1000 + i)
            out.append(r.randint(0, 256, (clip_len,) + tuple(like.shape[1:])).astype(like.dtype))
    return out
