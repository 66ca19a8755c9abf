"""Host side of the on-device CLIP image preprocessing.

`pil_bicubic_tables` restates Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (src/libImaging/Resample.c:
bicubic a = -0.5, support scaled by the down-scale factor, double arithmetic, 22-bit fixed point) and the size
rule of `CLIPImageProcessor` (shortest edge -> 224 with int() truncation of the long side, centre crop 224).
The resampling itself runs in HIP (`tspo_preprocess_frames`)."""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import Tuple

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_tables(in_size: int, out_size: int, first: int = 0, count: int = None) -> Tuple[np.ndarray, np.ndarray]:
    """int32 taps [count, ksize] and bounds [count, 2] (first input index, tap count) for output positions
    first .. first+count-1 of an in_size -> out_size resample.  out_size == in_size -> identity taps (Pillow skips
    the pass)."""
    count = out_size - first if count is None else count
    if out_size == in_size:
        kk = np.full((count, 1), 1 << PRECISION_BITS, np.int32)
        bounds = np.stack([np.arange(first, first + count), np.ones(count, np.int64)], 1).astype(np.int32)
        return kk, bounds
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((count, ksize), np.int32)
    bounds = np.zeros((count, 2), np.int32)
    ss = 1.0 / filterscale
    for i in range(count):
        xx = first + i
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            kk[i, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[i] = (xmin, xmax)
    return kk, bounds


def clip_resize_geometry(H: int, W: int, size: int = 224):
    """CLIPImageProcessor: shortest edge -> size (long side int(size * long / short)), then centre crop size x size."""
    short, long = (W, H) if W <= H else (H, W)
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if W <= H else (new_long, new_short)
    top, left = (new_h - size) // 2, (new_w - size) // 2
    return new_h, new_w, top, left


@functools.lru_cache(maxsize=32)
def _tables(H: int, W: int, size: int):
    new_h, new_w, top, left = clip_resize_geometry(H, W, size)
    hk, hb = pil_bicubic_tables(W, new_w, left, size)
    vk, vb = pil_bicubic_tables(H, new_h, top, size)
    ylo = int(vb[:, 0].min())
    yhi = int((vb[:, 0] + vb[:, 1]).max())
    return hk, hb, vk, vb, ylo, yhi - ylo


_dev_tables = {}


def preprocess_frames(frames: torch.Tensor, size: int = 224) -> torch.Tensor:
    """uint8 CUDA frames [T,H,W,3] or [T,3,H,W] -> uint8 [T,3,size,size] (resize + centre crop, Pillow-exact)."""
    from . import _lib, ops
    ops._need_gpu(frames)
    if frames.dtype != torch.uint8 or frames.ndim != 4:
        raise TypeError("preprocess_frames expects a uint8 tensor [T,H,W,3] or [T,3,H,W]")
    layout = 0 if frames.shape[-1] == 3 else 1
    if layout == 1 and frames.shape[1] != 3:
        raise ValueError(f"cannot infer the channel axis of {tuple(frames.shape)}")
    fr = frames.contiguous()
    T = fr.shape[0]
    H, W = (fr.shape[1], fr.shape[2]) if layout == 0 else (fr.shape[2], fr.shape[3])
    key = (H, W, size, fr.device)
    if key not in _dev_tables:
        hk, hb, vk, vb, ylo, nrows = _tables(H, W, size)
        _dev_tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(fr.device) for a in (hk, hb, vk, vb)) + (ylo, nrows)
    hk, hb, vk, vb, ylo, nrows = _dev_tables[key]
    out = torch.empty((T, 3, size, size), dtype=torch.uint8, device=fr.device)
    nws = _lib.lib().tspo_preprocess_workspace_bytes(T, nrows, size)
    ws = torch.empty((nws,), dtype=torch.uint8, device=fr.device)
    _lib.check(_lib.lib().tspo_preprocess_frames(
        C.c_void_p(fr.data_ptr()), layout, T, H, W, C.c_void_p(hk.data_ptr()), C.c_void_p(hb.data_ptr()), size, hk.shape[1],
        C.c_void_p(vk.data_ptr()), C.c_void_p(vb.data_ptr()), size, vk.shape[1], ylo, nrows, C.c_void_p(out.data_ptr()),
        C.c_void_p(ws.data_ptr()), ws.numel(), torch.cuda.current_stream().cuda_stream), "tspo_preprocess_frames")
    return out


def processor_is_default_clip(image_processor, size: int = 224) -> bool:
    """True when a (HF) CLIPImageProcessor is configured exactly like openai/clip-vit-large-patch14's, i.e. when the
    on-device path is equivalent to calling it."""
    g = lambda n, d=None: getattr(image_processor, n, d)
    sz, cs = g("size", {}), g("crop_size", {})
    mean, std = g("image_mean", []), g("image_std", [])
    try:
        return bool(g("do_resize") and g("do_center_crop") and g("do_rescale") and g("do_normalize")
                    and dict(sz).get("shortest_edge") == size and int(g("resample")) == 3
                    and dict(cs).get("height") == size and dict(cs).get("width") == size
                    and abs(g("rescale_factor") - 1 / 255) < 1e-12
                    and np.allclose(mean, [0.48145466, 0.4578275, 0.40821073]) and np.allclose(std, [0.26862954, 0.26130258, 0.27577711]))
    except Exception:
        return False
