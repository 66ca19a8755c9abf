"""K1: on-device CLIP image preprocessing.  CPU part: the oracle restatement and the host tap tables against
Pillow / transformers' CLIPImageProcessor (bit-exact).  GPU part: the HIP kernel against the oracle (bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import tspo_oracle as O
from tspo_amd import preprocess as P, synth

SIZES = [(360, 640), (480, 360), (224, 224), (300, 224), (200, 150), (225, 400), (1080, 1920), (230, 231)]


@pytest.mark.parametrize("hw", SIZES[:6], ids=[f"{h}x{w}" for h, w in SIZES[:6]])
def test_oracle_matches_pil_and_clip_processor(hw):
    from PIL import Image
    from transformers import CLIPImageProcessor
    H, W = hw
    frames = synth.uniform_u8((2, H, W, 3), 100 + H)
    new_h, new_w, top, left = P.clip_resize_geometry(H, W)
    for t in range(2):
        ref = np.array(Image.fromarray(frames[t]).resize((new_w, new_h), resample=Image.BICUBIC))
        np.testing.assert_array_equal(O.pil_bicubic_resize_u8(frames[t], new_w, new_h), ref)
    got = O.clip_preprocess_u8(frames)
    proc = CLIPImageProcessor()
    assert P.processor_is_default_clip(proc)
    pv = proc(images=[Image.fromarray(f) for f in frames], return_tensors="pt")["pixel_values"]
    mine = O.clip_normalize_pixels(torch.from_numpy(got))
    np.testing.assert_allclose(mine.numpy(), pv.numpy(), rtol=0, atol=2e-6)


def test_host_tables_reproduce_oracle():
    """numpy evaluation of the exact integer arithmetic the kernel performs, with the product's tap tables."""
    for H, W in SIZES:
        frames = synth.uniform_u8((1, H, W, 3), 7 + W)
        hk, hb, vk, vb, ylo, nrows = P._tables(H, W, 224)
        img = frames[0].astype(np.int64)
        tmp = np.zeros((nrows, 224, 3), np.int64)
        for x in range(224):
            acc = np.full((nrows, 3), 1 << 21, np.int64)
            for i in range(hb[x, 1]):
                acc += img[ylo:ylo + nrows, hb[x, 0] + i] * int(hk[x, i])
            tmp[:, x] = np.clip(acc >> 22, 0, 255)
        out = np.zeros((224, 224, 3), np.int64)
        for y in range(224):
            acc = np.full((224, 3), 1 << 21, np.int64)
            for i in range(vb[y, 1]):
                acc += tmp[vb[y, 0] - ylo + i] * int(vk[y, i])
            out[y] = np.clip(acc >> 22, 0, 255)
        np.testing.assert_array_equal(out.transpose(2, 0, 1).astype(np.uint8), O.clip_preprocess_u8(frames)[0])


def test_processor_config_detection():
    from transformers import CLIPImageProcessor
    assert P.processor_is_default_clip(CLIPImageProcessor())
    assert not P.processor_is_default_clip(CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}))
    assert not P.processor_is_default_clip(CLIPImageProcessor(resample=2))
    assert not P.processor_is_default_clip(object())


GPU_SIZES = SIZES + [(720, 1280), (433, 577), (224, 1000), (2160, 3840)]    # + 720p (bench geometry), odd pitch, wide crop, 4K


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["matrix", "lds"])
@pytest.mark.parametrize("hw", GPU_SIZES, ids=[f"{h}x{w}" for h, w in GPU_SIZES])
def test_gpu_preprocess_bit_exact(hw, path, monkeypatch):
    """Bit-exact against the Pillow-pinned oracle through BOTH horizontal passes the C ABI can take at out_w = 224: the
    matrix-pipe kernel (tspo_preprocess_frames_ex with the digit-split tap matrices, the default) and the LDS-staged scalar kernel
    (what tspo_preprocess_frames - no matrix tables - and spans beyond 4 K-blocks run)."""
    monkeypatch.setattr(P, "USE_MATRIX_PASS", path == "matrix")
    H, W = hw
    T = 3 if H * W < 2000 * 2000 else 1
    frames = synth.uniform_u8((T, H, W, 3), 55 + H + W)
    ref = O.clip_preprocess_u8(frames)
    got = P.preprocess_frames(torch.from_numpy(frames).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    chw = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 3, 1, 2))).cuda()     # qwen25vl-style [T,3,H,W]
    np.testing.assert_array_equal(P.preprocess_frames(chw).cpu().numpy(), ref)
    # a tensor that does not start on a 4-byte boundary (view into a larger buffer): the LDS-staged loads realign per row
    big = torch.zeros(frames.size + 8, dtype=torch.uint8, device="cuda")
    for off in (1, 3):
        view = big[off:off + frames.size].view(T, H, W, 3)
        view.copy_(torch.from_numpy(frames).cuda())
        assert view.data_ptr() % 4 == off
        np.testing.assert_array_equal(P.preprocess_frames(view).cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("hw,size", [((97, 131), 222), ((360, 640), 222), ((720, 1280), 98), ((300, 224), 222), ((222, 222), 222)],
                         ids=["97x131->222", "360x640->222", "720p->98", "300x224->222", "identity-222"])
def test_gpu_preprocess_plain_scalar_fallback(hw, size):
    """An output width that is not a multiple of 4 (dword-aligned rows are what the LDS-staged and matrix passes need) takes
    the byte-granular kernels of round 2 - the path for odd sizes and misaligned caller buffers: same bit-exactness, both
    input layouts, including a pure crop (identity resize)."""
    H, W = hw
    frames = synth.uniform_u8((2, H, W, 3), 91 + H + W + size)
    ref = O.clip_preprocess_u8(frames, size=size)
    np.testing.assert_array_equal(P.preprocess_frames(torch.from_numpy(frames).cuda(), size=size).cpu().numpy(), ref)
    chw = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 3, 1, 2))).cuda()
    np.testing.assert_array_equal(P.preprocess_frames(chw, size=size).cpu().numpy(), ref)


@pytest.mark.gpu
def test_gpu_preprocess_rejects_bad_input():
    from tspo_amd._lib import TspoHipError
    with pytest.raises(TspoHipError):
        P.preprocess_frames(torch.zeros(2, 8, 8, 3, dtype=torch.uint8))
    with pytest.raises(TypeError):
        P.preprocess_frames(torch.zeros(2, 8, 8, 3, device="cuda"))
