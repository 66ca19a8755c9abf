"""Drop-in for the reference's ``model/utils.py`` (same names / signatures).

gumbel_softmax runs on the GPU (tspo_gumbel_topk); generate_uniform_integers
and AKS_sampling are host code as in the reference (utils.py:10-16, 83-153:
the reference itself runs AKS in numpy on the CPU and moves the result to the
GPU afterwards) - AKS in an own work-list formulation, pinned to the
reference's outputs by the aks8 / aks16 golden vectors.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


def generate_uniform_integers(t, l):
    """model/utils.py:10-16."""
    if l <= 0:
        return []
    if l == 1:
        return [t]
    step = t / (l - 1)
    return [round(i * step) for i in range(l)]


class _LogSoftmaxOfLogits(torch.autograd.Function):
    """log(softmax(logits)) computed by the HIP sampler; backward is the softmax Jacobian so the
    reference's autograd-style loss (tspo_trainer.py:544,594-607) can be back-propagated unchanged.
    (The fused trainer does not use this: it calls tspo_pg_grad_logits.)"""

    @staticmethod
    def forward(ctx, logits, logp):
        ctx.save_for_backward(logp)
        return logp.clone()

    @staticmethod
    def backward(ctx, g):
        (logp,) = ctx.saved_tensors
        return g - torch.exp(logp) * g.sum(), None


def gumbel_softmax(logits, tau=1.0, sample_len=64, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None,
                   offset: int = 0):
    """Gumbel-top-k frame sampler (model/utils.py:69-80).

    logits [T,1] -> (top_k_indices int64 [k] ascending, probs [T], log_probs [T]).
    `noise` ([T] Gumbel(0,1) draws) may be injected for reproducibility; otherwise the kernel draws from
    Philox4x32-10 keyed by (seed, offset) - seed defaults to a fresh draw from torch's global generator so
    that torch.manual_seed() controls the rollout like it does for F.gumbel_softmax in the reference.
    """
    from . import ops
    T = logits.shape[0]
    if sample_len > T:
        raise RuntimeError("selected index k out of range")       # torch.topk, utils.py:73
    lg = logits.reshape(1, T)
    if noise is None and seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    n = None if noise is None else noise.reshape(1, 1, T)
    out = ops.gumbel_topk(lg.detach(), int(sample_len), 1, noise=n, seed=seed or 0, offset=offset, tau=float(tau),
                          want_probs=True)
    log_probs = out["logp"][0]
    if logits.requires_grad:
        log_probs = _LogSoftmaxOfLogits.apply(logits.reshape(T).float(), log_probs)
    return out["idx"][0, 0], out["probs"][0, 0].to(logits.dtype), log_probs.to(logits.dtype)


def group_features_by_cluster(features, cluster_indices):
    """model/utils.py:42-50 (kept for API completeness; bin-max runs in tspo_binmax)."""
    return [features[cluster_indices == c] for c in torch.unique(cluster_indices)]


def extract_clip_features(clip_model, clip_processor, video, text):
    """model/utils.py:18-35."""
    from .temporal_agent import extract_clip_features_impl
    return extract_clip_features_impl(clip_model, clip_processor, video, text, 'llava')


def meanstd(len_scores, dic_scores, n, fns, t1, t2, all_depth):
    """The partition step of adaptive keyframe sampling under the reference's name and calling convention
    (model/utils.py:83-126; `AKS_sampling` below does not go through it): segments `{"score": array, "depth": d}` with their
    frame-number lists in, the final segments out - a segment whose best `n` scores stand out from its mean by more than `t1`
    (and whose std exceeds `t2`) or that has been halved `all_depth` times is final, any other is halved.  Returned level by
    level (final segments of depth d before those of depth d + 1, each level in input order), as the reference's recursion
    concatenates them.  `len_scores` is unused there as well."""
    done_s, done_f = [], []
    level_s, level_f = list(dic_scores), list(fns)
    while level_s:
        next_s, next_f = [], []
        for seg, frames in zip(level_s, level_f):
            sc, depth = seg["score"], seg["depth"]
            best = sorted(range(len(sc)), key=lambda i: -sc[i])[:n]          # stable: equal scores keep the lower frame first
            if (np.mean([sc[i] for i in best]) - np.mean(sc) > t1 and np.std(sc) > t2) or depth >= all_depth:
                done_s.append(seg)
                done_f.append(frames)
            else:
                half = len(sc) // 2
                next_s += [dict(score=sc[:half], depth=depth + 1), dict(score=sc[half:], depth=depth + 1)]
                next_f += [frames[:half], frames[half:]]
        level_s, level_f = next_s, next_f
    return done_s, done_f


def AKS_sampling(score, max_num_frames, t1: float = 0.2, t2: float = -100.0, all_depth: int = 3):
    """Adaptive keyframe sampling, the 'aks' branch of inference_ts (contract: model/utils.py:83-153 with its constants
    t1 = 0.2, t2 = -100, depth 3): a clip whose best `max_num_frames` frames stand out from its mean by more than t1 (on
    min-max normalised scores) keeps its share of the budget as plain top-k; a flat clip is halved, down to `all_depth`
    halvings, each half owning half of the parent's budget.  Host numpy like the reference (it moves the indices to the
    GPU afterwards), but formulated over index ranges of ONE normalised array with an explicit work list instead of
    recursively copied score / frame-number lists.  Ties go to the lower frame number (the order `heapq.nlargest` over
    indices yields), and the means are taken over the values in that same order, so results match bit for bit."""
    print("t1", t1, " all_depth", all_depth)
    score = np.asarray(score)
    total, budget = len(score), int(max_num_frames)
    if total < budget:
        return list(range(total))
    z = (score - np.min(score)) / (np.max(score) - np.min(score))
    chosen = []
    work = [(0, total, 0)]                                   # (first frame, one past the last, halvings so far)
    while work:
        lo, hi, depth = work.pop()
        clip = z[lo:hi]
        if clip.size == 0:
            continue
        by_rank = np.argsort(-clip, kind="stable")           # value descending, equal values -> lower frame first
        stands_out = np.mean(clip[by_rank[:budget]]) - np.mean(clip) > t1 and np.std(clip) > t2
        if stands_out or depth >= all_depth:
            chosen.extend((lo + by_rank[: int(budget / 2 ** depth)]).tolist())
        else:
            mid = lo + (hi - lo) // 2
            work += [(lo, mid, depth + 1), (mid, hi, depth + 1)]
    chosen.sort()
    return chosen
