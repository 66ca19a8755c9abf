"""Drop-in for the reference's ``model/utils.py`` (same names / signatures).

gumbel_softmax runs on the GPU (tspo_gumbel_topk); generate_uniform_integers
and AKS_sampling are host code exactly as in the reference (utils.py:10-16,
83-153: the reference itself runs AKS in numpy on the CPU and moves the
result to the GPU afterwards).
"""
from __future__ import annotations

import heapq
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


def _meanstd(len_scores, dic_scores, n, fns, t1, t2, all_depth):
    """model/utils.py:83-130."""
    split_scores, split_fn, no_split_scores, no_split_fn = [], [], [], []
    for dic_score, fn in zip(dic_scores, fns):
        score, depth = dic_score['score'], dic_score['depth']
        mean, std = np.mean(score), np.std(score)
        top_n = heapq.nlargest(n, range(len(score)), score.__getitem__)
        mean_diff = np.mean([score[t] for t in top_n]) - mean
        if mean_diff > t1 and std > t2:
            no_split_scores.append(dic_score)
            no_split_fn.append(fn)
        elif depth < all_depth:
            half = len(score) // 2
            split_scores.append(dict(score=score[:half], depth=depth + 1))
            split_scores.append(dict(score=score[half:], depth=depth + 1))
            split_fn.append(fn[:half])
            split_fn.append(fn[half:])
        else:
            no_split_scores.append(dic_score)
            no_split_fn.append(fn)
    if len(split_scores) > 0:
        all_split_score, all_split_fn = _meanstd(len_scores, split_scores, n, split_fn, t1, t2, all_depth)
    else:
        all_split_score, all_split_fn = [], []
    return no_split_scores + all_split_score, no_split_fn + all_split_fn


meanstd = _meanstd


def AKS_sampling(score, max_num_frames):
    """model/utils.py:132-153 (t1=0.2, t2=-100, all_depth=3)."""
    t1, t2, all_depth = 0.2, -100, 3
    print("t1", t1, " all_depth", all_depth)
    fn = [x for x in range(len(score))]
    num = max_num_frames
    if len(score) >= num:
        normalized_data = (score - np.min(score)) / (np.max(score) - np.min(score))
        a, b = _meanstd(len(score), [dict(score=normalized_data, depth=0)], num, [fn], t1, t2, all_depth)
        out = []
        for s, f in zip(a, b):
            f_num = int(num / 2 ** (s['depth']))
            topk = heapq.nlargest(f_num, range(len(s['score'])), s['score'].__getitem__)
            out.extend([f[t] for t in topk])
        out.sort()
        return out
    return fn
