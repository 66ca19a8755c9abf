"""ctypes binding of libtspo_hip.so (include/tspo_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a
tensor is not on the GPU the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libtspo_hip.so")

TSPO_F32, TSPO_BF16, TSPO_F16, TSPO_U8 = 0, 1, 2, 3
TSPO_CLIP_NO_LN_FOLD, TSPO_CLIP_PRUNE_LAST, TSPO_CLIP_FOLD_CACHED = 1, 2, 4
ABI_VERSION = 4

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t
_u64 = C.c_uint64


class SelectorWeights(C.Structure):
    _fields_ = [(n, _p) for n in ("wqkv", "bqkv", "w1", "b1", "w2", "b2")]


class SelectorGrads(C.Structure):
    _fields_ = [(n, _p) for n in ("wqkv", "bqkv", "w1", "b1", "w2", "b2")]


class ClipConfig(C.Structure):
    _fields_ = [("hidden", _i), ("layers", _i), ("heads", _i), ("mlp", _i), ("patch", _i), ("image", _i),
                ("proj", _i), ("ln_eps", _f)]


class ClipLayer(C.Structure):
    _fields_ = [(n, _p) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2")]


class ClipWeights(C.Structure):
    _fields_ = [("cfg", ClipConfig), ("patch_w", _p), ("pos_emb", _p), ("pre_g", _p), ("pre_b", _p),
                ("post_g", _p), ("post_b", _p), ("proj_w", _p), ("layers", C.POINTER(ClipLayer))]


# name -> (restype, argtypes); every symbol include/tspo_hip.h declares
SIGNATURES = {
    "tspo_version": (_i, []),
    "tspo_last_error": (C.c_char_p, []),
    "tspo_topk_sorted": (_i, [_p, _i, _i, _i, _p, _p]),
    "tspo_binmax": (_i, [_p, _i, _i, _i, _p, _p]),
    "tspo_gumbel_topk": (_i, [_p, _p, _u64, _u64, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p]),
    "tspo_gumbel_topk_ex": (_i, [_p, _p, _u64, _u64, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _i]),
    "tspo_grpo_advantage": (_i, [_p, _i, _i, _f, _p, _p]),
    "tspo_pg_grad_logits": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p]),
    "tspo_grpo_pg_grad": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p, _p, _p]),
    "tspo_selector_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "tspo_selector_forward": (_i, [C.POINTER(SelectorWeights), _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _sz, _p]),
    "tspo_selector_backward": (_i, [C.POINTER(SelectorWeights), _p, _p, _p, _i, _i, _i, _i, _i, _i, _f,
                                    C.POINTER(SelectorGrads), _p, _sz, _p]),
    "tspo_selector_forward_ex": (_i, [C.POINTER(SelectorWeights), _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _sz, _p, _i]),
    "tspo_selector_backward_ex": (_i, [C.POINTER(SelectorWeights), _p, _p, _p, _i, _i, _i, _i, _i, _i, _f,
                                       C.POINTER(SelectorGrads), _p, _sz, _p, _i]),
    "tspo_policy_backward": (_i, [C.POINTER(SelectorWeights), _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _i, _f, _f,
                                  C.POINTER(SelectorGrads), _p, _p, _p, _sz, _p, _i]),
    "tspo_policy_backward_ex": (_i, [C.POINTER(SelectorWeights), _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _i, _f, _f,
                                     C.POINTER(SelectorGrads), _p, _p, _p, _sz, _p, _i, _p, C.POINTER(C.c_int)]),
    "tspo_grad_norm_scale": (_i, [_p, _sz, _f, _f, _p, _p, _sz, _p]),
    "tspo_adamw_step": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _i, _f, _p, _p]),
    "tspo_adamw_clip_step": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _i, _f, _f, _p, _p, _sz, _p]),
    "tspo_adamw_clip_step_ex": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _i, _f, _f, _p, _p, _i, _p]),
    "tspo_clip_workspace_bytes": (_sz, [C.POINTER(ClipConfig), _i]),
    "tspo_clip_vit_forward": (_i, [C.POINTER(ClipWeights), _p, _i, _i, _p, _p, _sz, _p]),
    "tspo_clip_vit_forward_ex": (_i, [C.POINTER(ClipWeights), _p, _i, _i, _p, _p, _sz, _p, _i]),
    "tspo_clip_vit_profile": (_i, [C.POINTER(ClipWeights), _p, _i, _i, _p, _p, _sz, _p, C.POINTER(C.c_float), _i]),
    "tspo_preprocess_workspace_bytes": (_sz, [_i, _i, _i]),
    "tspo_preprocess_frames": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "tspo_preprocess_frames_ex": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p, _p, _p, _p, _i, _i]),
    "tspo_clip_scores": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "tspo_gemm_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
}

_lib = None


class TspoHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TspoHipError(
            f"{LIB_PATH} is missing - build it with `python -m tspo_amd.build` (hipcc --offload-arch=gfx950). "
            "tspo_amd has no CPU / eager fallback.")
    import torch  # noqa: F401  - makes torch's bundled libamdhip64.so.7 the HIP runtime this process uses
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    if l.tspo_version() != ABI_VERSION:
        raise TspoHipError(f"ABI version mismatch: library {l.tspo_version()} != binding {ABI_VERSION}")
    _lib = l
    return l


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().tspo_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise TspoHipError(f"{what} failed (rc={rc}): {msg}")
