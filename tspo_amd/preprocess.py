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
USE_MATRIX_PASS = True      # horizontal pass on the matrix pipe (tspo_preprocess_frames_ex); False = scalar kernels (tests A/B both)


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


def mfma_h_tables(hk: np.ndarray, hb: np.ndarray):
    """Tables of the matrix-pipe horizontal pass (tspo_preprocess_frames_ex) from the tap table of the crop window:
    -> (taps int8 [nblk, nkb, 3, 64, 16], bias int32 [ow], xs int32 [nblk], nkb, span) or None when the window is not a
    whole number of 16-column blocks.  Block b covers output columns 16b..16b+15 and input columns xs[b] .. xs[b]+64*nkb-1;
    every 22-bit tap k is split into signed byte digits k = d0 + 256 d1 + 65536 d2 (|k| < 2^23, so d2 fits a byte too);
    lane l = q*16 + x of a fragment holds the 16 bytes of output column x for input columns q*16..q*16+15 of the K block."""
    ow, ksize = hk.shape
    if ow % 16 or ksize < 2:
        return None
    nblk = ow // 16
    xs = hb[::16, 0].astype(np.int64)
    last = hb[15::16]
    spans = last[:, 0] + last[:, 1] - xs
    nkb = int((spans.max() + 63) // 64)
    if nkb < 1 or nkb > 4:
        return None
    taps = np.zeros((nblk, nkb * 64, 16), np.int64)           # [block][input column within block][output column within block]
    for x in range(ow):
        b, xi = divmod(x, 16)
        o = int(hb[x, 0] - xs[b])
        taps[b, o:o + int(hb[x, 1]), xi] = hk[x, :int(hb[x, 1])]
    d0 = ((taps + 128) % 256) - 128
    t1 = (taps - d0) // 256
    d1 = ((t1 + 128) % 256) - 128
    d2 = (t1 - d1) // 256
    assert np.abs(d2).max() <= 127 and np.array_equal(d0 + 256 * d1 + 65536 * d2, taps)
    dig = np.stack([d0, d1, d2], 0).astype(np.int8)            # [digit][block][k][x]
    # -> [block][K block][digit][q][x][16 consecutive k]
    dig = dig.reshape(3, nblk, nkb, 4, 16, 16)                  # [d][b][kb][q][j][x]
    frag = np.ascontiguousarray(dig.transpose(1, 2, 0, 3, 5, 4)).reshape(nblk, nkb, 3, 64, 16)
    bias = (128 * hk.astype(np.int64).sum(1) + (1 << 21)).astype(np.int64)
    assert bias.max() < 2 ** 31
    # widest span of a chunk of four consecutive blocks (what one workgroup stages), in pixels
    span = max(int(xs[min(c + 3, nblk - 1)] + nkb * 64 - xs[c]) for c in range(0, nblk, 4))
    return frag, bias.astype(np.int32), xs.astype(np.int32), nkb, span


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
        mf = mfma_h_tables(hk, hb)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(fr.device)
        _dev_tables[key] = tuple(dev(a) for a in (hk, hb, vk, vb)) + (ylo, nrows) + \
            ((dev(mf[0]), dev(mf[1]), dev(mf[2]), mf[3], mf[4]) if mf is not None else (None, None, None, 0, 0))
    hk, hb, vk, vb, ylo, nrows, mtaps, mbias, mxs, mnkb, mspan = _dev_tables[key]
    out = torch.empty((T, 3, size, size), dtype=torch.uint8, device=fr.device)
    nws = _lib.lib().tspo_preprocess_workspace_bytes(T, nrows, size)
    ws = torch.empty((nws,), dtype=torch.uint8, device=fr.device)
    P = lambda t_: None if t_ is None else C.c_void_p(t_.data_ptr())
    _lib.check(_lib.lib().tspo_preprocess_frames_ex(
        C.c_void_p(fr.data_ptr()), layout, T, H, W, C.c_void_p(hk.data_ptr()), C.c_void_p(hb.data_ptr()), size, hk.shape[1],
        C.c_void_p(vk.data_ptr()), C.c_void_p(vb.data_ptr()), size, vk.shape[1], ylo, nrows, C.c_void_p(out.data_ptr()),
        C.c_void_p(ws.data_ptr()), ws.numel(), torch.cuda.current_stream().cuda_stream,
        P(mtaps) if USE_MATRIX_PASS else None, P(mbias), P(mxs), int(mnkb), int(mspan)), "tspo_preprocess_frames_ex")
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
