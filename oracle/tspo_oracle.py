"""CPU oracle for the TSPO temporal-sampling hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU fp32 / numpy, the arithmetic of the
reference (Hui-design/TSPO) for the one hot path this repository accelerates.
It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product
(``tspo_amd``) never imports it and has no CPU fallback.

Parity pinning: the reference has no tests and no golden vectors of its own
(SURVEY.md section 4, 8c).  The oracle is therefore pinned against outputs of the
reference *itself*, imported from /root/reference in the authoring container
by ``tests/golden/make_golden.py``; the resulting arrays are committed under
``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks every
function here against them.  CLIP-L arithmetic lives in third-party
``transformers`` (pinned ==4.49.0 by the reference, requirements.txt:322;
5.15.0 installed here) - the CLIP restatement is pinned against that
library's ``CLIPVisionModelWithProjection`` on closed-form weights.

Each function cites the reference lines it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import heapq
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# a2  positional_encoding            model/temporal_agent.py:10-19
# --------------------------------------------------------------------------

def positional_encoding(T: int, C: int) -> torch.Tensor:
    """Sinusoid table with the position normalised by T (temporal_agent.py:15)."""
    div_term = torch.exp(torch.arange(0, C, 2) * (-torch.log(torch.tensor(10000.0)) / C))
    pe = torch.zeros(1, T, C)
    position = torch.arange(T).unsqueeze(1) / T
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


# --------------------------------------------------------------------------
# a3  window mask                    model/temporal_agent.py:97-104
# --------------------------------------------------------------------------

def window_bounds(T: int, window_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """[lo, hi] (inclusive) of the key *set* row j attends to.

    The reference marks ``clamp(j - w//2 + k, 0, T-1)`` for k < w
    (temporal_agent.py:101-102); the clamped images of a contiguous range
    form the contiguous range [max(0, j-w//2), min(T-1, j-w//2+w-1)].
    """
    j = np.arange(T)
    lo = np.maximum(j - window_size // 2, 0)
    hi = np.minimum(j - window_size // 2 + window_size - 1, T - 1)
    return lo, hi


def create_window_mask(seq_len: int, window_size: int = 8) -> torch.Tensor:
    lo, hi = window_bounds(seq_len, window_size)
    cols = np.arange(seq_len)[None, :]
    m = (cols >= lo[:, None]) & (cols <= hi[:, None])
    return torch.from_numpy(m.astype(np.float32))


# --------------------------------------------------------------------------
# a4 + a5  selector forward          model/temporal_agent.py:38-79, 106-143
# --------------------------------------------------------------------------

SELECTOR_KEYS = (
    "temporal.Self_q.weight", "temporal.Self_q.bias",
    "temporal.Self_k.weight", "temporal.Self_k.bias",
    "temporal.Self_v.weight", "temporal.Self_v.bias",
    "temporal.ffn_o.weight", "temporal.ffn_o.bias",
    "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias",
)


def pair_cosine(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """temporal_agent.py:106-114 - eps is added to the *product* of norms."""
    dots = torch.einsum("bnc,bmc->bnm", a, b)
    a_norm = torch.sqrt((a ** 2).sum(dim=-1)).unsqueeze(-1)
    b_norm = torch.sqrt((b ** 2).sum(dim=-1)).unsqueeze(1)
    return dots / (a_norm * b_norm + 1e-6)


def selector_forward(params: Dict[str, torch.Tensor], input_emb: torch.Tensor,
                     text_emb: torch.Tensor, clip_scores: torch.Tensor,
                     window_size: int, score_tau: float = 0.025,
                     num_heads: int = 8) -> Tuple[torch.Tensor, torch.Tensor]:
    """MultiModal_Align.forward (temporal_agent.py:116-143).

    input_emb [T,D], text_emb [M,D] or [1,M,D], clip_scores [T]
    -> (sim_total [T], temporal_attn [1,T,D]).  ffn_o is never applied
    (temporal_agent.py:77-79); the global attention branch is multiplied by
    alpha = 0.0 (:54-55) so only the windowed softmax contributes.
    """
    T, D = input_emb.shape
    hd = D // num_heads
    x = input_emb.unsqueeze(0)
    mask = create_window_mask(T, window_size).to(x.dtype)
    x_t = x + positional_encoding(T, D).to(x.dtype)
    q = F.linear(x_t, params["temporal.Self_q.weight"], params["temporal.Self_q.bias"])
    k = F.linear(x_t, params["temporal.Self_k.weight"], params["temporal.Self_k.bias"])
    v = F.linear(x_t, params["temporal.Self_v.weight"], params["temporal.Self_v.bias"])
    q = q.view(1, T, num_heads, hd).permute(0, 2, 1, 3)
    k = k.view(1, T, num_heads, hd).permute(0, 2, 1, 3)
    v = v.view(1, T, num_heads, hd).permute(0, 2, 1, 3)
    scores = torch.matmul(q, k.transpose(-2, -1)) / (hd ** 0.5)
    scores = scores.masked_fill(mask == 0, -1e6)
    attn = torch.matmul(F.softmax(scores, dim=-1), v)
    ctx = attn.transpose(1, 2).contiguous().view(1, T, D)
    h = F.linear(F.relu(F.linear(ctx, params["mlp.0.weight"], params["mlp.0.bias"])),
                 params["mlp.2.weight"], params["mlp.2.bias"]) + x
    if text_emb.ndim == 2:
        text_emb = text_emb.unsqueeze(0)
    sim = pair_cosine(h, text_emb)[0].mean(dim=-1)
    sim = sim + clip_scores
    return sim / score_tau, h


# --------------------------------------------------------------------------
# a7  inference_ts                   model/temporal_agent.py:187-214
# --------------------------------------------------------------------------

def topk_sorted(confidence: torch.Tensor, sample_len: int) -> torch.Tensor:
    """'topk' branch (temporal_agent.py:190-192).

    torch.topk leaves the order among exactly-equal values unspecified; this
    oracle (and the HIP kernel) declare ties -> lowest index.
    """
    c = confidence.detach().float().cpu().numpy()
    k = min(len(c), sample_len)
    key = np.where(np.isnan(c), np.inf, c)
    order = np.lexsort((np.arange(len(c)), -key))  # value desc, index asc
    return torch.from_numpy(np.sort(order[:k]).astype(np.int64))


def generate_uniform_integers(t: int, l: int) -> List[int]:
    """model/utils.py:10-16 (Python round = half-to-even on the double)."""
    if l <= 0:
        return []
    if l == 1:
        return [t]
    step = t / (l - 1)
    return [round(i * step) for i in range(l)]


def binmax(confidence: torch.Tensor, sample_len: int) -> torch.Tensor:
    """'bin-max' branch (temporal_agent.py:194-210): nearest-anchor bins
    (first-min on ties), first arg-max per bin."""
    c = confidence.detach().float().cpu().numpy()
    T = len(c)
    k = min(T, sample_len)
    anchors = np.asarray(generate_uniform_integers(T - 1, k), dtype=np.int64)
    slots = np.array([int(np.argmin(np.abs(x - anchors))) for x in range(T)])
    out = []
    for s in np.unique(slots):
        members = np.nonzero(slots == s)[0]
        out.append(int(members[int(np.argmax(c[members]))]))
    return torch.tensor(out, dtype=torch.int64)


def _aks_meanstd(len_scores, dic_scores, n, fns, t1, t2, all_depth):
    """model/utils.py:83-130 (recursive split)."""
    split_scores, split_fn, no_split_scores, no_split_fn = [], [], [], []
    for dic_score, fn in zip(dic_scores, fns):
        score, depth = dic_score["score"], dic_score["depth"]
        mean, std = np.mean(score), np.std(score)
        top_n = heapq.nlargest(n, range(len(score)), score.__getitem__)
        mean_diff = np.mean([score[t] for t in top_n]) - mean
        if mean_diff > t1 and std > t2:
            no_split_scores.append(dic_score); no_split_fn.append(fn)
        elif depth < all_depth:
            h = len(score) // 2
            split_scores += [dict(score=score[:h], depth=depth + 1), dict(score=score[h:], depth=depth + 1)]
            split_fn += [fn[:h], fn[h:]]
        else:
            no_split_scores.append(dic_score); no_split_fn.append(fn)
    if split_scores:
        a, b = _aks_meanstd(len_scores, split_scores, n, split_fn, t1, t2, all_depth)
    else:
        a, b = [], []
    return no_split_scores + a, no_split_fn + b


def aks_sampling(score: np.ndarray, max_num_frames: int) -> List[int]:
    """model/utils.py:132-153 (t1=0.2, t2=-100, depth 3)."""
    t1, t2, all_depth = 0.2, -100, 3
    fn = list(range(len(score)))
    num = max_num_frames
    if len(score) < num:
        return fn
    normalized = (score - np.min(score)) / (np.max(score) - np.min(score))
    a, b = _aks_meanstd(len(score), [dict(score=normalized, depth=0)], num, [fn], t1, t2, all_depth)
    out = []
    for s, f in zip(a, b):
        f_num = int(num / 2 ** (s["depth"]))
        topk = heapq.nlargest(f_num, range(len(s["score"])), s["score"].__getitem__)
        out.extend(f[t] for t in topk)
    out.sort()
    return out


# --------------------------------------------------------------------------
# a8  Gumbel-top-k sampler           model/utils.py:69-80
# --------------------------------------------------------------------------

def gumbel_topk(logits: torch.Tensor, noise: torch.Tensor, sample_len: int,
                tau: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """gumbel_softmax with the Gumbel noise g = -log(Exp(1)) *injected*.

    logits [T], noise [T] -> (idx int64 [k] ascending, probs [T], log_probs [T]).
    F.gumbel_softmax computes softmax((logits + g) / tau) (utils.py:70);
    the top-k of that equals the top-k of logits + g wherever the softmax
    values are distinct; ties (incl. underflow to equal values) break towards
    the larger perturbed logit, then the lowest index.
    """
    if sample_len > logits.numel():
        raise RuntimeError("selected index k out of range")  # torch.topk, utils.py:73
    z = (logits.float() + noise.float()) / tau
    y = F.softmax(z, dim=0)
    idx = topk_sorted(z, sample_len)
    one_hot = torch.zeros_like(y).scatter_(0, idx, 1.0)
    probs = (one_hot - y) + y
    log_probs = F.softmax(logits.float(), dim=0).log()
    return idx, probs, log_probs


# -- counter-based RNG used by the HIP sampler when no noise is injected ----
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """Philox4x32-10 (Salmon et al. 2011).  ctr uint32 [...,4] -> uint32 [...,4]."""
    c = ctr.astype(np.uint64)
    k0, k1 = np.uint64(key[0] & 0xFFFFFFFF), np.uint64(key[1] & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c0
        p1 = np.uint64(PHILOX_M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & mask, lo1, (hi0 ^ c3 ^ k1) & mask, lo0
        k0 = (k0 + np.uint64(PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(PHILOX_W1)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def gumbel_noise_philox(B: int, G: int, T: int, seed: int, offset: int) -> np.ndarray:
    """The sampler's in-kernel noise stream, restated.

    Element (b,g,t): counter = (t, g, b, offset_lo), key = (seed_lo, seed_hi ^ offset_hi);
    u = (x0 + 0.5) * 2^-32 in (0,1); g = -log(-log(u)) evaluated in fp32.
    """
    b, g, t = np.meshgrid(np.arange(B), np.arange(G), np.arange(T), indexing="ij")
    ctr = np.stack([t, g, b, np.full_like(t, offset & 0xFFFFFFFF)], axis=-1).astype(np.uint32)
    key = (seed & 0xFFFFFFFF, ((seed >> 32) ^ (offset >> 32)) & 0xFFFFFFFF)
    x0 = philox4x32_10(ctr, key)[..., 0]
    u = ((x0.astype(np.float64) + 0.5) * (2.0 ** -32)).astype(np.float32)
    u = np.minimum(u, np.float32(1.0 - 2.0 ** -24))
    return (-np.log(-np.log(u))).astype(np.float32)


# --------------------------------------------------------------------------
# a9  training-time temporal_sampling   llava_qwen.py:131-144
# --------------------------------------------------------------------------

def policy_temporal_sampling(params, image_embeddings, text_features, clip_scores, noise,
                             sample_len=64, ts_ids=None, window_size=None, score_tau=0.025):
    confidence, _ = selector_forward(params, image_embeddings, text_features, clip_scores,
                                     window_size, score_tau)
    assert confidence.ndim == 1
    idx, _probs, logp = gumbel_topk(confidence, noise, sample_len)
    sel = (idx.clone(), idx) if ts_ids is None else ts_ids
    return sel, logp, confidence


# --------------------------------------------------------------------------
# a10  group-relative advantage      tspo_trainer.py:587-592
# --------------------------------------------------------------------------

def grpo_advantage(rewards: torch.Tensor, num_generations: int, eps: float = 1e-4) -> torch.Tensor:
    """rewards [B*G] -> advantages [B*G]; unbiased std over the G rollouts."""
    r = rewards.view(-1, num_generations)
    mean = r.mean(dim=1).repeat_interleave(num_generations, dim=0)
    std = r.std(dim=1).repeat_interleave(num_generations, dim=0)
    return (rewards - mean) / (std + eps)


# --------------------------------------------------------------------------
# a11  policy-gradient loss          tspo_trainer.py:594-609 (+ :544)
# --------------------------------------------------------------------------

def pg_loss(ts_logps: Sequence[torch.Tensor], advantages: torch.Tensor) -> torch.Tensor:
    """L = -(1/G) sum_g A_g * mean_j exp(lp_gj - sg(lp_gj))  (autograd form)."""
    G = len(ts_logps)
    total = 0.0
    for g in range(G):
        item = torch.exp(ts_logps[g] - ts_logps[g].detach()).mean()
        total = total + (-(item * advantages[g]))
    return total / G


def pg_grad_logits(logits: torch.Tensor, idx: torch.Tensor, adv: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Closed form of dL/dlogits for one prompt.

    logits [T], idx int64 [G,k], adv [G] -> (loss scalar = -mean(A), dL/ds [T]):
    dL/ds_t = -(1/G) sum_g A_g (1[t in S_g]/k - softmax(s)_t).
    """
    G, k = idx.shape
    p = F.softmax(logits.float(), dim=0)
    grad = torch.zeros_like(p)
    for g in range(G):
        m = torch.zeros_like(p)
        m[idx[g]] = 1.0 / k
        grad += -(adv[g] / G) * (m - p)
    return -adv.mean(), grad


def tspo_step_autograd(params: Dict[str, torch.Tensor], image, text, clip, noise, rewards,
                       k: int, window_size: int, tau: float):
    """Reference-shaped training step for ONE prompt (tspo_trainer.py:500-609):
    G rollouts, then G re-evaluations with grad, advantage, PG loss, backward.

    noise [G,T], rewards [G].  Returns (idx [G,k], loss, {name: grad}).
    """
    G = noise.shape[0]
    ps = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    with torch.no_grad():
        idxs = [policy_temporal_sampling(ps, image, text, clip, noise[g], k, None, window_size, tau)[0]
                for g in range(G)]
    logps = []
    for g in range(G):
        _, lp, _ = policy_temporal_sampling(ps, image, text, clip, noise[g], k, idxs[g], window_size, tau)
        logps.append(lp[idxs[g][1]])
    adv = grpo_advantage(rewards, G)
    loss = pg_loss(logps, adv)
    loss.backward()
    grads = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in ps.items()}
    return torch.stack([i[1] for i in idxs]), loss.detach(), grads


def tspo_step_autograd_one_forward(params: Dict[str, torch.Tensor], image, text, clip, noise, rewards,
                                   k: int, window_size: int, tau: float, num_heads: int = 8):
    """tspo_step_autograd with the G re-evaluations collapsed into ONE differentiable forward.

    The reference re-runs the selector once per rollout (tspo_trainer.py:540-552) on unchanged weights and inputs, so the
    G confidence vectors are the same tensor G times and the loss (:594-607) is a sum of G gathers from it; autograd's
    gradient of that sum is the same whether the G copies are separate graphs or one.  This form exists for LONG videos
    (T = 4096: a dense T x T graph is ~2 GB, G = 16 of them do not fit a test box) and is checked against
    tspo_step_autograd itself in tests/test_oracle_golden.py.  Returns (idx [G,k], loss, {name: grad}, adv [G], scores [T]).
    """
    G = noise.shape[0]
    ps = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    conf, _ = selector_forward(ps, image, text, clip, window_size, tau, num_heads)
    with torch.no_grad():
        idxs = [gumbel_topk(conf.detach(), noise[g], k)[0] for g in range(G)]
    logp = F.softmax(conf, dim=0).log()                       # model/utils.py:78
    adv = grpo_advantage(rewards, G)
    loss = pg_loss([logp[i] for i in idxs], adv)
    loss.backward()
    grads = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in ps.items()}
    return torch.stack(idxs), loss.detach(), grads, adv, conf.detach()


# --------------------------------------------------------------------------
# K16  AdamW (torch.optim.AdamW semantics, HF Trainer default)
# --------------------------------------------------------------------------

def adamw_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0,
               grad_scale: float = 1.0):
    g = g * grad_scale
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def clip_grad_scale(total_norm: float, max_norm: float = 1.0) -> float:
    """torch.nn.utils.clip_grad_norm_ coefficient (clamped to 1)."""
    return min(1.0, max_norm / (total_norm + 1e-6))


# --------------------------------------------------------------------------
# a1  CLIP ViT vision tower + projection  (transformers CLIPVisionTransformer;
#     reference call sites temporal_agent.py:166, tspo_trainer.py:401)
# --------------------------------------------------------------------------

def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def clip_vit_forward(w: Dict[str, torch.Tensor], pixel_values: torch.Tensor, *, num_heads: int,
                     patch: int, eps: float = 1e-5, return_hidden: bool = False):
    """CLIPModel.get_image_features restated (pre-LN ViT, quick_gelu, CLS pool,
    post-LN, bias-free projection).  ``w`` uses the HF state-dict key names of
    ``CLIPVisionModelWithProjection`` (``vision_model.*`` / ``visual_projection.weight``).
    pixel_values [N,3,H,W] fp32 -> features [N, proj_dim].
    """
    pre = "vision_model."
    x = F.conv2d(pixel_values, w[pre + "embeddings.patch_embedding.weight"], stride=patch)
    N, C = x.shape[0], x.shape[1]
    x = x.flatten(2).transpose(1, 2)
    cls = w[pre + "embeddings.class_embedding"].expand(N, 1, C)
    x = torch.cat([cls, x], dim=1) + w[pre + "embeddings.position_embedding.weight"].unsqueeze(0)
    x = F.layer_norm(x, (C,), w[pre + "pre_layrnorm.weight"], w[pre + "pre_layrnorm.bias"], eps)
    hidden = [x]
    L = 1 + max(int(k.split(".")[3]) for k in w if k.startswith(pre + "encoder.layers."))
    hd = C // num_heads
    S = x.shape[1]
    for l in range(L):
        p = f"{pre}encoder.layers.{l}."
        h = F.layer_norm(x, (C,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], eps)
        q = F.linear(h, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"])
        k = F.linear(h, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"])
        v = F.linear(h, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"])
        q = q.view(N, S, num_heads, hd).transpose(1, 2)
        k = k.view(N, S, num_heads, hd).transpose(1, 2)
        v = v.view(N, S, num_heads, hd).transpose(1, 2)
        a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        a = torch.matmul(a, v).transpose(1, 2).reshape(N, S, C)
        x = x + F.linear(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (C,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], eps)
        h = quick_gelu(F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        hidden.append(x)
    pooled = F.layer_norm(x[:, 0, :], (C,), w[pre + "post_layernorm.weight"], w[pre + "post_layernorm.bias"], eps)
    feats = F.linear(pooled, w["visual_projection.weight"])
    return (feats, hidden) if return_hidden else feats


def clip_cosine_scores(text_features: torch.Tensor, image_features: torch.Tensor) -> torch.Tensor:
    """torch.nn.CosineSimilarity(dim=-1)(text [1,D], image [T,D]) (temporal_agent.py:167)."""
    return F.cosine_similarity(text_features.float(), image_features.float(), dim=-1)


def clip_normalize_pixels(u8: torch.Tensor) -> torch.Tensor:
    """CLIPImageProcessor rescale + normalise on already 224x224 frames
    ([N,3,H,W] uint8 -> fp32): (x/255 - mean)/std."""
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    return (u8.float() / 255.0 - mean) / std


# --------------------------------------------------------------------------
# end-to-end helpers used by bench.py's cpu_baseline
# --------------------------------------------------------------------------

def frames_scored_path(clip_w, sel_params, pixels, text_features, *, num_heads, patch, k, window_size=12,
                       tau=0.025):
    """pixels -> CLIP features -> clip score -> selector -> greedy top-k."""
    feats = clip_vit_forward(clip_w, pixels, num_heads=num_heads, patch=patch)
    clip = clip_cosine_scores(text_features, feats)
    scores, _ = selector_forward(sel_params, feats, text_features, clip, window_size, tau)
    return topk_sorted(scores, k), scores, feats


# --------------------------------------------------------------------------
# K1  CLIPImageProcessor front half: PIL antialiased bicubic resize + centre crop
#     (third-party: Pillow src/libImaging/Resample.c, transformers CLIPImageProcessor;
#      reference call sites model/temporal_agent.py:156-164, tspo_trainer.py:393-399)
# --------------------------------------------------------------------------

def _pil_bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _pil_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc (PRECISION_BITS = 22)."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 2.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int64)
    bounds = np.zeros((out_size, 2), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_pil_bicubic((x + xmin - center + 0.5) * (1.0 / fs)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            v = v / ww if ww != 0.0 else v
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _pil_resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    kk, b = _pil_coeffs(a.shape[0], out_size)
    out = np.zeros((out_size,) + a.shape[1:], np.int64)
    for xx in range(out_size):
        xmin, xmax = b[xx]
        acc = np.full(a.shape[1:], 1 << 21, np.int64)
        for x in range(xmax):
            acc += a[xmin + x] * kk[xx, x]
        out[xx] = np.clip(acc >> 22, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def pil_bicubic_resize_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """uint8 [H,W,C] -> [new_h,new_w,C], bit-identical to PIL.Image.resize(..., BICUBIC): horizontal pass into a
    rounded uint8 image, then vertical pass; a pass whose size does not change is skipped."""
    h, w = img.shape[:2]
    out = img
    if new_w != w:
        out = _pil_resample_axis(out, new_w, 1)
    if new_h != h:
        out = _pil_resample_axis(out, new_h, 0)
    return out


def clip_preprocess_u8(frames: np.ndarray, size: int = 224) -> np.ndarray:
    """uint8 [T,H,W,3] -> uint8 [T,3,size,size]: CLIPImageProcessor's resize(shortest_edge) + center_crop
    (rescale / normalise are applied afterwards: clip_normalize_pixels)."""
    T, H, W, _ = frames.shape
    short, long = (W, H) if W <= H else (H, W)
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if W <= H else (new_long, new_short)
    top, left = (new_h - size) // 2, (new_w - size) // 2
    out = np.zeros((T, 3, size, size), np.uint8)
    for t in range(T):
        r = pil_bicubic_resize_u8(frames[t], new_w, new_h)
        out[t] = r[top:top + size, left:left + size].transpose(2, 0, 1)
    return out
