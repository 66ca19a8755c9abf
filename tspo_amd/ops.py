"""Tensor-level wrappers over the C ABI (include/tspo_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every
computation below is a hand-written HIP kernel in libtspo_hip.so.  All
functions require CUDA(ROCm) tensors and raise otherwise - there is no CPU
path in the product.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import TSPO_BF16, TSPO_F16, TSPO_F32, TSPO_U8, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.TspoHipError("tspo_amd ops need GPU tensors (no CPU fallback); got a tensor on " + str(t.device))


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


import os as _os

DEBUG_CHECKS = _os.environ.get("TSPO_DEBUG_CHECKS", "0") not in ("", "0")   # costs a host sync per call: off by default


def _check_rollout_idx(ix: torch.Tensor, what: str, need_group: bool = False) -> None:
    """The policy-gradient kernels find a frame's rollout membership by BINARY SEARCH in idx[b,g,:]: every rollout's
    index list must be strictly ascending (what tspo_gumbel_topk / tspo_topk_sorted emit).  An unsorted external list (a
    reference-style `ts_ids` in selection order) would give wrong gradients silently, so with DEBUG_CHECKS (or
    TSPO_DEBUG_CHECKS=1) it is rejected here.  A group of one rollout has no spread (std of one value = NaN, as
    torch.std gives): rejected where an advantage is formed."""
    if need_group and ix.shape[1] < 2:
        raise ValueError(f"{what}: G = {ix.shape[1]} rollout per prompt - the group-relative advantage needs G >= 2")
    if DEBUG_CHECKS and ix.shape[-1] > 1 and not bool((ix[..., 1:] > ix[..., :-1]).all()):
        raise ValueError(f"{what}: idx[b,g,:] must be strictly ascending (sort each rollout's frame list first)")


# ---------------------------------------------------------------------------
# samplers
# ---------------------------------------------------------------------------

def topk_sorted(scores: torch.Tensor, k: int) -> torch.Tensor:
    """scores [T] or [B,T] -> ascending int64 indices [min(T,k)] / [B,min(T,k)]."""
    _need_gpu(scores)
    squeeze = scores.ndim == 1
    s = _f32c(scores.view(1, -1) if squeeze else scores)
    B, T = s.shape
    ke = min(T, int(k))
    if ke < 1:
        raise ValueError("topk_sorted: k must be >= 1")
    idx = torch.empty((B, ke), dtype=torch.int64, device=s.device)
    check(_lib.lib().tspo_topk_sorted(_ptr(s), B, T, int(k), _ptr(idx), _stream()), "tspo_topk_sorted")
    return idx[0] if squeeze else idx


def binmax(scores: torch.Tensor, k: int) -> torch.Tensor:
    _need_gpu(scores)
    squeeze = scores.ndim == 1
    s = _f32c(scores.view(1, -1) if squeeze else scores)
    B, T = s.shape
    ke = min(T, int(k))
    if ke < 1:
        raise ValueError("binmax: k must be >= 1")
    idx = torch.empty((B, ke), dtype=torch.int64, device=s.device)
    check(_lib.lib().tspo_binmax(_ptr(s), B, T, int(k), _ptr(idx), _stream()), "tspo_binmax")
    return idx[0] if squeeze else idx


def gumbel_topk(logits: torch.Tensor, k: int, G: int = 1, noise: Optional[torch.Tensor] = None, seed: int = 0,
                offset: int = 0, tau: float = 1.0, want_probs: bool = False, want_noise: bool = False,
                prompts_per_offset: int = 0):
    """logits [B,T]; noise [B,G,T] or None (in-kernel Philox).  Returns dict with idx [B,G,k], logp [B,T]
    and optionally probs / noise [B,G,T].  prompts_per_offset = p > 0: the batch holds B / p micro-steps of p prompts;
    group j draws what a separate call with offset + j would (coalesced micro-steps, bitwise the sequential rollouts)."""
    _need_gpu(logits, noise)
    l = _f32c(logits)
    B, T = l.shape
    if k > T:
        raise RuntimeError(f"selected index k out of range (k={k} > T={T})")  # torch.topk's error class
    n = None
    if noise is not None:
        n = _f32c(noise)
        if tuple(n.shape) != (B, G, T):
            raise ValueError(f"noise must be [B,G,T]={B, G, T}, got {tuple(n.shape)}")
    dev = l.device
    idx = torch.empty((B, G, k), dtype=torch.int64, device=dev)
    logp = torch.empty((B, T), dtype=torch.float32, device=dev)
    probs = torch.empty((B, G, T), dtype=torch.float32, device=dev) if want_probs else None
    nout = torch.empty((B, G, T), dtype=torch.float32, device=dev) if want_noise else None
    if prompts_per_offset < 0 or (prompts_per_offset and B % prompts_per_offset):
        raise ValueError(f"prompts_per_offset={prompts_per_offset} does not divide B={B}")
    check(_lib.lib().tspo_gumbel_topk_ex(_ptr(l), _ptr(n), seed & (2 ** 64 - 1), offset & (2 ** 64 - 1), B, G, T, int(k),
                                         float(tau), _ptr(idx), _ptr(logp), _ptr(probs), _ptr(nout), _stream(),
                                         int(prompts_per_offset)),
          "tspo_gumbel_topk")
    return {"idx": idx, "logp": logp, "probs": probs, "noise": nout}


def grpo_advantage(rewards: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """rewards [B,G] -> advantages [B,G]."""
    _need_gpu(rewards)
    r = _f32c(rewards)
    B, G = r.shape
    adv = torch.empty_like(r)
    check(_lib.lib().tspo_grpo_advantage(_ptr(r), B, G, float(eps), _ptr(adv), _stream()), "tspo_grpo_advantage")
    return adv


def pg_grad_logits(logp: torch.Tensor, idx: torch.Tensor, adv: torch.Tensor, scale: float = 1.0):
    """logp [B,T], idx [B,G,k] int64 ascending, adv [B,G] -> (dlogits [B,T], loss [B])."""
    _need_gpu(logp, idx, adv)
    lp, a = _f32c(logp), _f32c(adv)
    ix = idx.to(torch.int64).contiguous()
    B, T = lp.shape
    _, G, k = ix.shape
    _check_rollout_idx(ix, "pg_grad_logits")
    dl = torch.empty_like(lp)
    loss = torch.empty((B,), dtype=torch.float32, device=lp.device)
    check(_lib.lib().tspo_pg_grad_logits(_ptr(lp), _ptr(ix), _ptr(a), B, G, T, k, float(scale), _ptr(dl), _ptr(loss),
                                         _stream()), "tspo_pg_grad_logits")
    return dl, loss


def grpo_pg_grad(rewards: torch.Tensor, logp: torch.Tensor, idx: torch.Tensor, scale: float = 1.0, eps: float = 1e-4):
    """grpo_advantage + pg_grad_logits in one launch: rewards [B,G], logp [B,T], idx [B,G,k] ->
    (adv [B,G], dlogits [B,T], loss [B]); identical results."""
    _need_gpu(rewards, logp, idx)
    r, lp = _f32c(rewards), _f32c(logp)
    ix = idx.to(torch.int64).contiguous()
    B, T = lp.shape
    _, G, k = ix.shape
    if tuple(r.shape) != (B, G):
        raise ValueError(f"rewards {tuple(r.shape)} do not match idx {tuple(ix.shape)}")
    _check_rollout_idx(ix, "grpo_pg_grad", need_group=True)
    adv = torch.empty_like(r)
    dl = torch.empty_like(lp)
    loss = torch.empty((B,), dtype=torch.float32, device=lp.device)
    check(_lib.lib().tspo_grpo_pg_grad(_ptr(r), _ptr(lp), _ptr(ix), B, G, T, k, float(eps), float(scale), _ptr(adv), _ptr(dl),
                                       _ptr(loss), _stream()), "tspo_grpo_pg_grad")
    return adv, dl, loss


# ---------------------------------------------------------------------------
# selector (flat parameter bucket; reference key names are views into it)
# ---------------------------------------------------------------------------
# order inside the flat bucket; q|k|v weights are contiguous so the fused [3D,D] projection needs no copy
FLAT_LAYOUT = (
    ("temporal.Self_q.weight", "w"), ("temporal.Self_k.weight", "w"), ("temporal.Self_v.weight", "w"),
    ("temporal.Self_q.bias", "b"), ("temporal.Self_k.bias", "b"), ("temporal.Self_v.bias", "b"),
    ("mlp.0.weight", "w"), ("mlp.0.bias", "b"), ("mlp.2.weight", "w"), ("mlp.2.bias", "b"),
    ("temporal.ffn_o.weight", "w"), ("temporal.ffn_o.bias", "b"),   # present, never used (temporal_agent.py:77-79)
)


def flat_offsets(D: int) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
    off, out = 0, {}
    for name, kind in FLAT_LAYOUT:
        shape = (D, D) if kind == "w" else (D,)
        out[name] = (off, shape)
        off += D * D if kind == "w" else D
    out["__total__"] = (off, ())
    return out


def trainable_numel(D: int) -> int:
    """elements of the bucket that receive gradients (everything except ffn_o)."""
    return flat_offsets(D)["temporal.ffn_o.weight"][0]


def _sel_structs(flat: torch.Tensor, D: int, cls):
    o = flat_offsets(D)
    base = flat.data_ptr()
    s = cls()
    s.wqkv = base + 4 * o["temporal.Self_q.weight"][0]
    s.bqkv = base + 4 * o["temporal.Self_q.bias"][0]
    s.w1 = base + 4 * o["mlp.0.weight"][0]
    s.b1 = base + 4 * o["mlp.0.bias"][0]
    s.w2 = base + 4 * o["mlp.2.weight"][0]
    s.b2 = base + 4 * o["mlp.2.bias"][0]
    return s


def selector_workspace(B, T, D, H, M, window, device) -> torch.Tensor:
    n = _lib.lib().tspo_selector_workspace_bytes(B, T, D, H, M, window)
    if n == 0:
        raise ValueError("selector: bad dims")
    return torch.empty((n,), dtype=torch.uint8, device=device)


SEL_BF16X3 = 1       # include/tspo_hip.h: TSPO_SEL_BF16X3
SEL_ACCUMULATE = 2   # include/tspo_hip.h: TSPO_SEL_ACCUMULATE (backward calls: add to the gradient buffers)
SEL_BF16 = 4         # include/tspo_hip.h: TSPO_SEL_BF16 (forward only: bf16 GEMM operands, fp32 accumulate - the reference's inference precision)


def _sel_flags(precision: str, accumulate: bool = False, forward: bool = False) -> int:
    acc = SEL_ACCUMULATE if accumulate else 0
    if precision == "fp32":
        return acc
    if precision == "bf16x3":
        return SEL_BF16X3 | acc
    if precision == "bf16" and forward:
        return SEL_BF16
    raise ValueError(f"selector precision must be 'fp32' or 'bf16x3' (forward / inference also 'bf16'), got {precision!r}")


def selector_forward(flat: torch.Tensor, img: torch.Tensor, txt: torch.Tensor, clip: Optional[torch.Tensor],
                     H: int, window: int, tau: float, want_attn: bool = True, ws: Optional[torch.Tensor] = None,
                     precision: str = "fp32"):
    """flat: f32 parameter bucket (FLAT_LAYOUT). img [B,T,D], txt [B,M,D], clip [B,T] ->
    (scores [B,T] f32, temporal_attn [B,T,D] f32 | None, workspace).
    precision="bf16x3" (opt-in, training): split-precision GEMMs on the bf16 MFMA, ~1e-5 relative error.
    precision="bf16" (opt-in, inference): the reference's own inference precision (gen_id_tspo.py:55) - GEMM operands rounded to
    bf16, fp32 accumulation, fp32 between the GEMMs; a workspace written by it is NOT valid for a backward."""
    _need_gpu(flat, img, txt, clip)
    assert flat.dtype == torch.float32 and flat.is_contiguous()
    x, e = _f32c(img), _f32c(txt)
    B, T, D = x.shape
    M = e.shape[1]
    c = _f32c(clip) if clip is not None else None
    if ws is None:
        ws = selector_workspace(B, T, D, H, M, window, x.device)
    scores = torch.empty((B, T), dtype=torch.float32, device=x.device)
    attn = torch.empty((B, T, D), dtype=torch.float32, device=x.device) if want_attn else None
    w = _sel_structs(flat, D, _lib.SelectorWeights)
    check(_lib.lib().tspo_selector_forward_ex(C.byref(w), _ptr(x), _ptr(e), _ptr(c), B, T, D, H, M, int(window), float(tau),
                                              _ptr(scores), _ptr(attn), _ptr(ws), ws.numel(), _stream(),
                                              _sel_flags(precision, forward=True)), "tspo_selector_forward")
    return scores, attn, ws


def selector_backward(flat: torch.Tensor, flat_grad: torch.Tensor, img, txt, dscores, H, window, tau, ws,
                      precision: str = "fp32", accumulate: bool = False):
    """Writes the gradient of every trainable tensor into `flat_grad` (same layout as `flat`); accumulate=True ADDS it to what
    `flat_grad` holds (gradient accumulation without a scratch bucket)."""
    _need_gpu(flat, flat_grad, img, txt, dscores, ws)
    x, e, d = _f32c(img), _f32c(txt), _f32c(dscores)
    B, T, D = x.shape
    M = e.shape[1]
    w = _sel_structs(flat, D, _lib.SelectorWeights)
    g = _sel_structs(flat_grad, D, _lib.SelectorGrads)
    check(_lib.lib().tspo_selector_backward_ex(C.byref(w), _ptr(x), _ptr(e), _ptr(d), B, T, D, H, M, int(window), float(tau),
                                               C.byref(g), _ptr(ws), ws.numel(), _stream(), _sel_flags(precision, accumulate)),
          "tspo_selector_backward")


def policy_backward(flat: torch.Tensor, flat_grad: torch.Tensor, img, txt, rewards, logp, idx, H, window, tau, ws,
                    scale: float = 1.0, eps: float = 1e-4, precision: str = "fp32", norm_partials: Optional[torch.Tensor] = None,
                    accumulate: bool = False):
    """grpo_pg_grad + selector_backward with one launch less (the score-gradient kernel derives dL/dscores from the
    rollouts itself): -> (adv [B,G], loss [B]); gradients go to `flat_grad` (equal to the two-call form to rounding).
    norm_partials (f32 [>=2048]): the backward's last kernel also leaves the gradient's partial sums of squares there
    and the call returns (adv, loss, n_partials) - feed them to adamw_clip_step(norm_partials=...) (one launch).
    accumulate=True ADDS the gradients to `flat_grad` (second.. micro-step of an accumulation window)."""
    _need_gpu(flat, flat_grad, img, txt, rewards, logp, idx, ws)
    x, e, r, lp = _f32c(img), _f32c(txt), _f32c(rewards), _f32c(logp)
    ix = idx.to(torch.int64).contiguous()
    B, T, D = x.shape
    M = e.shape[1]
    _, G, k = ix.shape
    if tuple(r.shape) != (B, G) or tuple(lp.shape) != (B, T):
        raise ValueError(f"rewards {tuple(r.shape)} / logp {tuple(lp.shape)} do not match idx {tuple(ix.shape)}, feats {tuple(x.shape)}")
    _check_rollout_idx(ix, "policy_backward", need_group=True)
    w = _sel_structs(flat, D, _lib.SelectorWeights)
    g = _sel_structs(flat_grad, D, _lib.SelectorGrads)
    adv = torch.empty_like(r)
    loss = torch.empty((B,), dtype=torch.float32, device=x.device)
    if norm_partials is not None:
        _need_gpu(norm_partials)
        if norm_partials.dtype != torch.float32 or norm_partials.numel() < 2048:
            raise ValueError("norm_partials must be a float32 tensor of >= 2048 elements")
        npart = C.c_int(0)
        check(_lib.lib().tspo_policy_backward_ex(C.byref(w), _ptr(x), _ptr(e), _ptr(r), _ptr(lp), _ptr(ix), B, T, D, H, M,
                                                 int(window), float(tau), G, k, float(eps), float(scale), C.byref(g), _ptr(adv),
                                                 _ptr(loss), _ptr(ws), ws.numel(), _stream(), _sel_flags(precision, accumulate),
                                                 _ptr(norm_partials), C.byref(npart)), "tspo_policy_backward_ex")
        return adv, loss, int(npart.value)
    check(_lib.lib().tspo_policy_backward(C.byref(w), _ptr(x), _ptr(e), _ptr(r), _ptr(lp), _ptr(ix), B, T, D, H, M, int(window),
                                          float(tau), G, k, float(eps), float(scale), C.byref(g), _ptr(adv), _ptr(loss), _ptr(ws),
                                          ws.numel(), _stream(), _sel_flags(precision, accumulate)), "tspo_policy_backward")
    return adv, loss


def grad_norm_scale(grad: torch.Tensor, n: int, pre_scale: float = 1.0, max_norm: float = 1.0,
                    out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> device tensor [2] = (||g||, clip coefficient * pre_scale); no host sync.  `out` / `ws` (>= 2048 bytes) let a
    caller that runs this every step keep its own buffers instead of two allocations per call."""
    _need_gpu(grad, out, ws)
    if out is None:
        out = torch.empty((2,), dtype=torch.float32, device=grad.device)
    if ws is None:
        ws = torch.empty((2048,), dtype=torch.uint8, device=grad.device)
    check(_lib.lib().tspo_grad_norm_scale(_ptr(grad), n, float(pre_scale), float(max_norm), _ptr(out), _ptr(ws),
                                          ws.numel(), _stream()), "tspo_grad_norm_scale")
    return out


def adamw_step(param, grad, m, v, n: int, lr: float, step: int, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
               grad_scale: float = 1.0, d_grad_scale: Optional[torch.Tensor] = None):
    _need_gpu(param, grad, m, v, d_grad_scale)
    check(_lib.lib().tspo_adamw_step(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), n, float(lr), float(beta1), float(beta2),
                                     float(eps), float(weight_decay), int(step), float(grad_scale), _ptr(d_grad_scale),
                                     _stream()), "tspo_adamw_step")


def adamw_clip_step(param, grad, m, v, n: int, lr: float, step: int, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                    pre_scale: float = 1.0, max_norm: float = 1.0, out: Optional[torch.Tensor] = None,
                    ws: Optional[torch.Tensor] = None, norm_partials: Optional[torch.Tensor] = None,
                    n_partials: int = 0) -> torch.Tensor:
    """grad_norm_scale + adamw_step (two launches instead of three) -> device tensor [2] = (||g||, applied scale).
    With norm_partials / n_partials from policy_backward(norm_partials=...): one launch."""
    _need_gpu(param, grad, m, v, out, ws, norm_partials)
    if out is None:
        out = torch.empty((2,), dtype=torch.float32, device=grad.device)
    if norm_partials is not None:
        check(_lib.lib().tspo_adamw_clip_step_ex(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), n, float(lr), float(beta1),
                                                 float(beta2), float(eps), float(weight_decay), int(step), float(pre_scale),
                                                 float(max_norm), _ptr(out), _ptr(norm_partials), int(n_partials), _stream()),
              "tspo_adamw_clip_step_ex")
        return out
    if ws is None:
        ws = torch.empty((2048,), dtype=torch.uint8, device=grad.device)
    check(_lib.lib().tspo_adamw_clip_step(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), n, float(lr), float(beta1),
                                          float(beta2), float(eps), float(weight_decay), int(step), float(pre_scale),
                                          float(max_norm), _ptr(out), _ptr(ws), ws.numel(), _stream()),
          "tspo_adamw_clip_step")
    return out


# ---------------------------------------------------------------------------
# CLIP ViT frame encoder
# ---------------------------------------------------------------------------
_PIX_DTYPES = {torch.float32: TSPO_F32, torch.bfloat16: TSPO_BF16, torch.float16: TSPO_F16, torch.uint8: TSPO_U8}


class ClipVitWeights:
    """Device-resident, kernel-ready copy of a HF CLIP vision tower + visual_projection
    (bf16 [out,in] matrices, fp32 vectors, q|k|v fused, class_embedding folded into pos row 0,
    conv weight flattened and K-padded to a multiple of 64)."""

    def __init__(self, state: Dict[str, torch.Tensor], cfg: dict, device):
        self.cfg = dict(cfg)
        dev = torch.device(device)
        self._keep = []

        def f32(t):
            t = torch.as_tensor(t).detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t

        def b16(t):
            t = torch.as_tensor(t).detach().to(device=dev, dtype=torch.float32).to(torch.bfloat16).contiguous()
            self._keep.append(t)
            return t

        p = "vision_model."
        C_, patch = cfg["hidden"], cfg["patch"]
        kreal = 3 * patch * patch
        kp = (kreal + 63) // 64 * 64
        pw = torch.as_tensor(state[p + "embeddings.patch_embedding.weight"]).float().reshape(C_, kreal)
        pw = torch.nn.functional.pad(pw, (0, kp - kreal))
        pos = torch.as_tensor(state[p + "embeddings.position_embedding.weight"]).float().clone()
        pos[0] += torch.as_tensor(state[p + "embeddings.class_embedding"]).float()
        self.layers = (_lib.ClipLayer * max(1, cfg["layers"]))()
        for l in range(cfg["layers"]):
            q = f"{p}encoder.layers.{l}."
            L = self.layers[l]
            wqkv = torch.cat([torch.as_tensor(state[q + f"self_attn.{n}_proj.weight"]).float() for n in "qkv"], 0)
            bqkv = torch.cat([torch.as_tensor(state[q + f"self_attn.{n}_proj.bias"]).float() for n in "qkv"], 0)
            L.ln1_g = f32(state[q + "layer_norm1.weight"]).data_ptr(); L.ln1_b = f32(state[q + "layer_norm1.bias"]).data_ptr()
            L.wqkv = b16(wqkv).data_ptr(); L.bqkv = f32(bqkv).data_ptr()
            L.wo = b16(state[q + "self_attn.out_proj.weight"]).data_ptr(); L.bo = f32(state[q + "self_attn.out_proj.bias"]).data_ptr()
            L.ln2_g = f32(state[q + "layer_norm2.weight"]).data_ptr(); L.ln2_b = f32(state[q + "layer_norm2.bias"]).data_ptr()
            L.w1 = b16(state[q + "mlp.fc1.weight"]).data_ptr(); L.b1 = f32(state[q + "mlp.fc1.bias"]).data_ptr()
            L.w2 = b16(state[q + "mlp.fc2.weight"]).data_ptr(); L.b2 = f32(state[q + "mlp.fc2.bias"]).data_ptr()
        w = _lib.ClipWeights()
        w.cfg = _lib.ClipConfig(cfg["hidden"], cfg["layers"], cfg["heads"], cfg["mlp"], cfg["patch"], cfg["image"],
                                cfg["proj"], float(cfg.get("ln_eps", 1e-5)))
        w.patch_w = b16(pw).data_ptr()
        w.pos_emb = f32(pos).data_ptr()
        w.pre_g = f32(state[p + "pre_layrnorm.weight"]).data_ptr(); w.pre_b = f32(state[p + "pre_layrnorm.bias"]).data_ptr()
        w.post_g = f32(state[p + "post_layernorm.weight"]).data_ptr(); w.post_b = f32(state[p + "post_layernorm.bias"]).data_ptr()
        w.proj_w = b16(state["visual_projection.weight"]).data_ptr()
        w.layers = C.cast(self.layers, C.POINTER(_lib.ClipLayer))
        self.struct = w
        self.device = dev
        self._ws = None
        self._fold_key = None   # (workspace, batch size, stream) the folded weights kept in the workspace are valid for

    def invalidate_fold_cache(self) -> None:
        """Forget the LayerNorm-folded weights kept in the workspace (needed by code that edits the packed weight tensors in place
        - nothing in this package does; writing to the buffer `workspace()` hands out invalidates by itself)."""
        self._fold_key = None

    def _workspace(self, n_frames: int) -> torch.Tensor:
        """The encoder's workspace as the LIBRARY uses it (caller-owned memory of the C ABI): the LayerNorm-folded weights of all
        layers live at its front (0.35 GB for CLIP-L/14, offsets independent of the batch), the activations behind them."""
        n = _lib.lib().tspo_clip_workspace_bytes(C.byref(self.struct.cfg), n_frames)
        if n == 0:
            raise ValueError("clip: bad config")
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty((n,), dtype=torch.uint8, device=self.device)
            self._fold_key = None
        return self._ws

    def workspace(self, n_frames: int) -> torch.Tensor:
        """The raw workspace buffer, for inspection (tests read the activations / statistics a forward left in it).  Handing it
        out ends the library's exclusive use of it, so the kept folded weights are no longer trusted: the next encode folds again
        (an external write to the front of the buffer must not silently yield wrong features)."""
        ws = self._workspace(n_frames)
        self._fold_key = None
        return ws

    def library_folds(self, n_frames: int) -> bool:
        """Whether an encode of n_frames takes the LayerNorm-folded path at all (mirrors csrc/clip_vit.hip: both folded GEMMs must be
        "big" problems, csrc/gemm_bf16.hip gemm_bf16_is_big - checked against the library's behaviour by tests/test_gpu_ops.py)."""
        c = self.cfg
        S = (c["image"] // c["patch"]) ** 2 + 1
        M = n_frames * S

        def big(N, K):
            return M * N >= 256 ** 3 and K >= 128 and N <= 4096
        return c["layers"] > 0 and big(c["hidden"], c["hidden"]) and big(c["hidden"], c["mlp"])


def _clip_flags(fold_layernorm: bool, prune_last_layer: bool, fold_cached: bool = False) -> int:
    return ((0 if fold_layernorm else _lib.TSPO_CLIP_NO_LN_FOLD) | (_lib.TSPO_CLIP_PRUNE_LAST if prune_last_layer else 0) |
            (_lib.TSPO_CLIP_FOLD_CACHED if fold_cached and fold_layernorm else 0))


def _fold_key(w, ws: torch.Tensor, n_frames: int):
    """What the folded weights kept in a ClipVitWeights workspace are valid FOR: this buffer and the stream they were written on -
    NOT the batch size (the folded weights do not depend on it: a video encoded in chunks with a shorter last chunk folds once);
    None when a batch of n_frames does not fold at all (nothing is written, nothing may be assumed).  The weights of a
    ClipVitWeights never change (a new state dict makes a new object, and with it a new workspace)."""
    if not w.library_folds(n_frames):
        return None
    return (ws.data_ptr(), ws.numel(), torch.cuda.current_stream(ws.device).cuda_stream)


def clip_vit_forward(w: ClipVitWeights, pixels: torch.Tensor, out: Optional[torch.Tensor] = None,
                     fold_layernorm: bool = True, prune_last_layer: bool = False) -> torch.Tensor:
    """pixels [N,3,H,W] (f32/bf16/f16 normalised, or uint8 raw) -> features f32 [N, proj].
    fold_layernorm=False keeps the stand-alone LayerNorm passes on large batches too (A/B comparison; small batches
    never fold).  prune_last_layer=True (opt-in) evaluates the last transformer block for the class-token row only -
    the other token rows of that block have no consumer (get_image_features pools row 0), so the features are the same;
    off by default so that the default path executes the full model like the reference does."""
    _need_gpu(pixels)
    if pixels.dtype not in _PIX_DTYPES:
        raise TypeError(f"unsupported pixel dtype {pixels.dtype}")
    px = pixels.contiguous()
    N = px.shape[0]
    cfg = w.cfg
    if tuple(px.shape[1:]) != (3, cfg["image"], cfg["image"]):
        raise ValueError(f"pixels must be [N,3,{cfg['image']},{cfg['image']}], got {tuple(px.shape)}")
    ws = w._workspace(N)
    feat = out if out is not None else torch.empty((N, cfg["proj"]), dtype=torch.float32, device=px.device)
    # the LayerNorm-folded weights of all layers stay in the workspace between calls (TSPO_CLIP_FOLD_CACHED): the second and later
    # encodes of a frozen CLIP tower on the same workspace skip the 2 x layers fold launches
    key = _fold_key(w, ws, N)
    cached = fold_layernorm and key is not None and w._fold_key == key
    check(_lib.lib().tspo_clip_vit_forward_ex(C.byref(w.struct), _ptr(px), _PIX_DTYPES[px.dtype], N, _ptr(feat), _ptr(ws),
                                              ws.numel(), _stream(), _clip_flags(fold_layernorm, prune_last_layer, cached)),
          "tspo_clip_vit_forward")
    if fold_layernorm and key is not None:
        w._fold_key = key
    return feat


def clip_vit_profile(w: ClipVitWeights, pixels: torch.Tensor, fold_layernorm: bool = True) -> Dict[str, float]:
    """One encode with a hipEvent after every launch -> ms per kernel class (profiling; synchronises)."""
    _need_gpu(pixels)
    px = pixels.contiguous()
    N = px.shape[0]
    ws = w._workspace(N)
    feat = torch.empty((N, w.cfg["proj"]), dtype=torch.float32, device=px.device)
    ms = (C.c_float * 6)()
    key = _fold_key(w, ws, N)
    cached = fold_layernorm and key is not None and w._fold_key == key
    check(_lib.lib().tspo_clip_vit_profile(C.byref(w.struct), _ptr(px), _PIX_DTYPES[px.dtype], N, _ptr(feat), _ptr(ws),
                                           ws.numel(), _stream(), ms, _clip_flags(fold_layernorm, False, cached)),
          "tspo_clip_vit_profile")
    if fold_layernorm and key is not None:
        w._fold_key = key
    return {"gemm_ms": ms[0], "attn_ms": ms[1], "ln_ms": ms[2], "gather_ms": ms[3], "total_ms": ms[4],
            "gemm_launches": int(ms[5])}


def clip_scores(txt: torch.Tensor, feat: torch.Tensor) -> torch.Tensor:
    """txt [B,M,D] (row 0 used), feat [B,T,D] -> cosine [B,T]."""
    _need_gpu(txt, feat)
    e, f = _f32c(txt), _f32c(feat)
    B, T, D = f.shape
    out = torch.empty((B, T), dtype=torch.float32, device=f.device)
    check(_lib.lib().tspo_clip_scores(_ptr(e), _ptr(f), B, T, D, e.shape[1], _ptr(out), _stream()), "tspo_clip_scores")
    return out


def gemm_bf16(A: torch.Tensor, W: torch.Tensor, bias=None, residual=None, act: int = 0, out_f32: bool = False,
              out: Optional[torch.Tensor] = None):
    """C = A @ W^T (+bias)(+residual | quick_gelu): the encoder's MFMA GEMM, exposed for tests / microbenchmarks.
    out: write into this [M, N] tensor; it may BE the residual (the encoder's out-proj / fc2 update the residual stream in place)."""
    _need_gpu(A, W, bias, residual)
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16
    A, W = A.contiguous(), W.contiguous()
    M, K = A.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=A.device)
    elif out.shape != (M, N) or out.dtype != (torch.float32 if out_f32 else torch.bfloat16) or not out.is_contiguous():
        raise ValueError("gemm_bf16: out must be a contiguous [M, N] tensor of the output dtype")
    b = _f32c(bias) if bias is not None else None
    r = residual.contiguous() if residual is not None else None
    check(_lib.lib().tspo_gemm_bf16(_ptr(A), _ptr(W), _ptr(b), _ptr(r), _ptr(out), TSPO_F32 if out_f32 else TSPO_BF16,
                                    M, N, K, int(act), _stream()), "tspo_gemm_bf16")
    return out
