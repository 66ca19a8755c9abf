"""Deterministic synthetic tensors (no RNG stream dependence).

Every element is a pure function of (seed, flat index): splitmix64 hash ->
two uniforms -> Box-Muller.  Used for synthetic CLIP / selector weights and
inputs in tests, golden-vector generation and bench.py so that the GPU box
and the authoring container see bit-identical fp32 inputs without shipping
any large fixture.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(shape, seed: int) -> np.ndarray:
    """fp64 uniforms in (0,1), element i = hash(seed, i)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(idx ^ _splitmix64(np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(1)))
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(shape, seed: int, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """fp32 N(mean, std) via Box-Muller on two hashed uniform streams."""
    u1 = uniform(shape, 2 * seed + 1)
    u2 = uniform(shape, 2 * seed + 2)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def uniform_u8(shape, seed: int) -> np.ndarray:
    return np.minimum(np.floor(uniform(shape, seed) * 256.0), 255).astype(np.uint8).reshape(shape)


def selector_state(dim: int = 768, seed: int = 7, std: float = 0.02, bias_std: float = 0.0) -> dict:
    """MultiModal_Align state dict (12 tensors; reference keys).
    HF init for a freshly created module is N(0, 0.02) weights / zero bias
    (tspo_trainer.py:201 via Qwen2 _init_weights); tests use bias_std > 0 to
    exercise the bias paths."""
    names = ["temporal.Self_q", "temporal.Self_k", "temporal.Self_v", "temporal.ffn_o", "mlp.0", "mlp.2"]
    out = {}
    for i, n in enumerate(names):
        out[n + ".weight"] = normal((dim, dim), seed * 100 + 2 * i, std)
        out[n + ".bias"] = (normal((dim,), seed * 100 + 2 * i + 1, bias_std) if bias_std > 0
                            else np.zeros((dim,), np.float32))
    return out


def clip_vision_state(hidden: int, layers: int, heads: int, mlp: int, patch: int, image: int,
                      proj: int, seed: int = 11, std: float = 0.02) -> dict:
    """HF `CLIPVisionModelWithProjection` state dict with synthetic weights."""
    s = [seed * 1000]

    def nxt():
        s[0] += 1
        return s[0]

    n_pos = (image // patch) ** 2 + 1
    p = "vision_model."
    w = {
        p + "embeddings.class_embedding": normal((hidden,), nxt(), std),
        p + "embeddings.patch_embedding.weight": normal((hidden, 3, patch, patch), nxt(), std),
        p + "embeddings.position_embedding.weight": normal((n_pos, hidden), nxt(), std),
        p + "pre_layrnorm.weight": normal((hidden,), nxt(), 0.05, 1.0),
        p + "pre_layrnorm.bias": normal((hidden,), nxt(), 0.02),
        p + "post_layernorm.weight": normal((hidden,), nxt(), 0.05, 1.0),
        p + "post_layernorm.bias": normal((hidden,), nxt(), 0.02),
        "visual_projection.weight": normal((proj, hidden), nxt(), std),
    }
    for l in range(layers):
        q = f"{p}encoder.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[q + f"self_attn.{nm}.weight"] = normal((hidden, hidden), nxt(), std)
            w[q + f"self_attn.{nm}.bias"] = normal((hidden,), nxt(), 0.02)
        for nm in ("layer_norm1", "layer_norm2"):
            w[q + nm + ".weight"] = normal((hidden,), nxt(), 0.05, 1.0)
            w[q + nm + ".bias"] = normal((hidden,), nxt(), 0.02)
        w[q + "mlp.fc1.weight"] = normal((mlp, hidden), nxt(), std)
        w[q + "mlp.fc1.bias"] = normal((mlp,), nxt(), 0.02)
        w[q + "mlp.fc2.weight"] = normal((hidden, mlp), nxt(), std)
        w[q + "mlp.fc2.bias"] = normal((hidden,), nxt(), 0.02)
    return w


def clip_vision_state_heavy_tailed(cfg: dict, seed: int = 23, outlier_channels=(7, 300, 911), outlier_gain: float = 50.0,
                                   gamma_hi: float = 10.0, row_mean: float = 2.0) -> dict:
    """clip_vision_state with the statistics real ViT-L/14 checkpoints have and smooth N(0, sigma) weights lack:
    a few residual-stream channels carry values ~50x the rest (the out-proj / fc2 rows and biases that write them are
    scaled), LayerNorm gains reach 10 on some channels and fall to 0.1 on others, and the residual rows are far from
    zero-mean (a constant added through the fc2 / out-proj biases).  This is where a LayerNorm folded into bf16 GEMMs
    (W' = bf16(gamma o W), statistics from the stored bf16 stream) would lose accuracy if it were going to."""
    w = clip_vision_state(**cfg, seed=seed)
    hid = cfg["hidden"]
    oc = np.array([c for c in outlier_channels if c < hid])
    hi = np.arange(5, hid, 97)          # ~1 % of the channels get gamma x10
    lo = np.arange(11, hid, 53)         # ~2 % get gamma x0.1
    p = "vision_model."
    w[p + "embeddings.position_embedding.weight"][:, oc] *= outlier_gain
    for l in range(cfg["layers"]):
        q = f"{p}encoder.layers.{l}."
        for nm in ("self_attn.out_proj", "mlp.fc2"):
            w[q + nm + ".weight"][oc, :] *= outlier_gain / 4
            w[q + nm + ".bias"] = w[q + nm + ".bias"] + np.float32(row_mean / (2 * cfg["layers"]))
            w[q + nm + ".bias"][oc] *= outlier_gain
        for nm in ("layer_norm1", "layer_norm2"):
            w[q + nm + ".weight"][hi] *= gamma_hi
            w[q + nm + ".weight"][lo] *= 0.1
    return w


CLIP_L14 = dict(hidden=1024, layers=24, heads=16, mlp=4096, patch=14, image=224, proj=768)
CLIP_TINY = dict(hidden=64, layers=2, heads=4, mlp=128, patch=14, image=28, proj=32)
