"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/tspo_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "tspo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tspo_[a-z0-9_]+)\s*\(", txt)))


def test_build_and_symbols():
    import __graft_entry__ as ge
    ge.build()
    from tspo_amd import _lib
    l = _lib.lib()
    hs = header_symbols()
    assert len(hs) >= 17
    for s in hs:
        assert hasattr(l, s), f"{s} declared in tspo_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == hs, "ctypes binding and header disagree"
    assert l.tspo_version() == 4 == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    """Entry points validate before touching the device: bad dims -> TSPO_EINVAL + message, no crash."""
    from tspo_amd import _lib
    l = _lib.lib()
    assert l.tspo_topk_sorted(None, 1, 8, 2, None, None) == -1
    assert b"null pointer" in l.tspo_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert l.tspo_gumbel_topk(p, None, 0, 0, 1, 1, 4, 5, 1.0, p, None, None, None, None) == -1   # k > T
    assert b"out of range" in l.tspo_last_error()
    assert l.tspo_selector_workspace_bytes(4, 512, 768, 8, 1, 12) > 0
    assert l.tspo_selector_workspace_bytes(0, 512, 768, 8, 1, 12) == 0
    cfg = _lib.ClipConfig(1024, 24, 16, 4096, 14, 224, 768, 1e-5)
    n1 = l.tspo_clip_workspace_bytes(ctypes.byref(cfg), 1)
    n64 = l.tspo_clip_workspace_bytes(ctypes.byref(cfg), 64)
    assert 0 < n1 < n64


def test_product_has_no_cpu_fallback():
    """ops refuse CPU tensors, and the package never imports the oracle."""
    import torch
    import pytest
    from tspo_amd import ops, _lib
    with pytest.raises(_lib.TspoHipError):
        ops.topk_sorted(torch.zeros(8), 2)
    with pytest.raises(_lib.TspoHipError):
        ops.grpo_advantage(torch.zeros(2, 4))
    pkg = os.path.join(ROOT, "tspo_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_shipped_library_has_no_dev_hooks():
    """The product sources carry no conditional development code (round 4: the laboratory lives in csrc/dev/*.hip, linked in only
    by `python -m tspo_amd.build --dev` and reached through a weak symbol): no TSPO_DEV_HOOKS / lab macro anywhere under csrc/
    outside dev/, the shipped library rejects the lab variant numbers before touching the device, does not export the lab entry
    points, holds none of the retired kernels and reads no environment variable."""
    import glob
    from tspo_amd import _lib, build as b
    for f in glob.glob(os.path.join(b.CSRC, "*.hip")) + glob.glob(os.path.join(b.CSRC, "*.h")):
        txt = open(f).read()
        for marker in ("TSPO_DEV_HOOKS", "TSPO_A9_LAB", "TSPO_ATTN_NO_SGB", "getenv"):
            assert marker not in txt, (f, marker)
    assert all(s.startswith("dev/") for s in b.DEV_SOURCES) and not any(s.startswith("dev/") for s in b.SOURCES)
    l = _lib.lib()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert b.built_mode() == "product", "the library on disk is a --dev build (python -m tspo_amd.build rebuilds the shipped one)"
    for variant in (2, 6, 8, 9, 60, 67, 69, 70, 71, 72, 74, 75, 76, 78, 82, 85):   # retired kernels / ablations / probes, the DMA-kernel lab schedules, the round-2 register-staged kernel
        assert l.tspo_gemm_bf16(p, p, p, None, p, _lib.TSPO_BF16, 4096, 4096, 1024, variant << 8, None) == -1
        assert b"not part of this build" in l.tspo_last_error(), (variant, l.tspo_last_error())
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"gemm_bf16_p3_kernel", b"gemm_bf16_s256_kernel", b"gemm_bf16_w16_kernel", b"gemm_bf16_p256_kernel", b"clip_attn257p_kernel",
                 b"clip_attn257w8_kernel", b"TSPO_GEMM_VARIANT", b"TSPO_ATTN_ABL", b"TSPO_SEL_SPLIT", b"getenv", b"tspo_dma_set_debug",
                 b"tspo_lab_gemm_dma\0tspo_dev", b"gemm_bf16_a7_kernel", b"gemm_bf16_a9lab_kernel"):
        assert name not in blob, name
    for sym in ("tspo_dma_set_debug", "tspo_dev_set_debug"):
        assert not hasattr(l, sym), sym
    assert b"gemm_bf16_a9_kernel" in blob


# SGPR spills (v_writelane / v_readlane into a spare VGPR, no memory traffic) hipcc leaves in each epilogue form of the production GEMM:
# none in the steady-state K-step of any form; the counts below are prologue / tile-switch / epilogue / remainder-phase code.  An
# upper bound per form, so that DESIGN.md cannot drift from the binary again.
A9_SGPR_SPILL_BOUND = {0: 0, 1: 0, 2: 24, 3: 0, 4: 12, 5: 4, 6: 4, 7: 72, 8: 0}


def test_agpr_gemm_code_audit(tmp_path):
    """gemm_dma.hip (production) keeps 256 accumulators per lane in AGPRs under literal names that the compiler does not know
    about.  That is only sound if hipcc itself never touches an AGPR in the kernel and does not spill inside the MFMA loop: audit
    the generated gfx950 code of every instantiation (device-only -S compile, ~1-2 min).  Also: the only m0 writes are the ones in
    front of its own DMA instructions; no scratch at all; no VGPR spills; SGPR spills bounded per form and absent from the
    steady-state K-step; and the same register checks for the encoder's attention kernel."""
    import subprocess
    from tspo_amd import build as b
    src, kernel = "gemm_dma.hip", "gemm_bf16_a9_kernel"
    asm = tmp_path / (src + ".s")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           os.path.join(b.CSRC, src), "-o", str(asm)])
    txt = open(asm).read()
    kernels = re.findall(r"^(_ZN\S*%s\S*):[^\n]*\n(.*?)\n\.Lfunc_end" % kernel, txt, flags=re.S | re.M)
    assert len(kernels) == 9, [k[0] for k in kernels]   # the eight epilogue forms + the head-major q/k/v form (GE_BIAS_LN_HM)
    meta = {m.group(1): m.group(2) for m in re.finditer(r"\.name:\s+(\S*%s\S*)\n(.*?)\.wavefront_size" % kernel, txt, flags=re.S)}
    for name, body in kernels:
        epi = int(re.search(r"kernelILi(\d+)E", name).group(1))
        inasm, bad, m0bad = False, [], []
        blocks, cur = [], []
        for ln in body.split("\n"):
            t = ln.strip()
            if re.match(r"^\.LBB\d+_\d+:", t):
                blocks.append(cur)
                cur = []
            if t.startswith(";;#ASMSTART"):
                inasm = True
            elif t.startswith(";;#ASMEND"):
                inasm = False
            elif not inasm and t and not t.startswith((";", ".")) and (re.search(r"\ba\[?\d+", t) or "accvgpr" in t):
                bad.append(t)
            elif not inasm and t and not t.startswith((";", ".")) and re.search(r"\bm0\b", t):
                m0bad.append(t)
            cur.append(t)
        blocks.append(cur)
        assert not bad, f"{name}: compiler-emitted AGPR access outside the asm statements: {bad[:3]}"
        # round 5: nothing interlocks a VALU write with the operand read of an MFMA the compiler cannot see inside an asm statement (it
        # may assemble an operand tuple with v_movs directly in front of the statement: garbage row statistics until found) - every
        # hand-written MFMA is either preceded by an s_nop or its A / B operands are not written by the two instructions before it
        ins = [t for t in (ln.strip() for ln in body.split("\n")) if t and not t.startswith((";", ".")) and not re.match(r"^\.?LBB", t)]
        def regs(tok):
            m = re.match(r"v\[(\d+):(\d+)\]", tok)
            if m:
                return set(range(int(m.group(1)), int(m.group(2)) + 1))
            m = re.match(r"v(\d+)$", tok)
            return {int(m.group(1))} if m else set()
        for i, t in enumerate(ins):
            if not t.startswith("v_mfma"):
                continue
            ops_ = [x.strip() for x in t.split(None, 1)[1].split(",")]
            src = regs(ops_[1]) | regs(ops_[2])
            for prev in reversed(ins[max(0, i - 2):i]):      # nearest first; an s_nop in between settles it
                if prev.startswith("s_nop"):
                    break
                if prev.startswith("v_") and not prev.startswith("v_mfma"):
                    dst = regs(prev.split(None, 1)[1].split(",")[0].strip())
                    assert not (dst & src), f"{name}: VALU write of an MFMA operand right in front of it: {prev} -> {t}"
        assert not m0bad, f"{name}: compiler-emitted m0 use next to the hand-written LDS-DMA: {m0bad[:3]}"
        assert not any(x.startswith("scratch_") for blk in blocks for x in blk), f"{name}: scratch access"
        kblocks = [blk for blk in blocks if sum(x.startswith("v_mfma") for x in blk) >= 100]      # the K-step bodies (the epilogue of the
        # residual forms holds MFMAs too since round 5 - residual rows and row statistics on the matrix pipe - in blocks of at most 72)
        assert len(kblocks) == 3                                                                  # first / steady-state / last K-step of a tile
        for blk in kblocks:
            assert not any(x.startswith("scratch_") for x in blk), f"{name}: scratch access inside an MFMA block"
        # the steady-state K-step = the second of the three unrolled K-step bodies (first of a tile / middle / last): no SGPR spill
        # traffic in its straight-line part, i.e. before its first branch (what follows is the tile switch, taken once per tile)
        steady = kblocks[1]
        upto = next((i for i, x in enumerate(steady) if x.startswith("s_cbranch")), len(steady))
        assert sum(x.startswith("v_mfma") for x in steady[:upto]) == 128, name
        assert not any("v_readlane" in x or "v_writelane" in x for x in steady[:upto]), f"{name}: SGPR spill traffic inside the steady-state K-step"
        assert body.count("v_mfma_f32_16x16x32_bf16") >= 3 * 128 + 8
        md = meta[name]
        val = lambda key: int(re.search(key + r":\s+(\d+)", md).group(1))
        assert val(r"\.vgpr_spill_count") == 0 and val(r"\.private_segment_fixed_size") == 0, name
        assert val(r"\.sgpr_spill_count") <= A9_SGPR_SPILL_BOUND[epi], (name, val(r"\.sgpr_spill_count"))


def test_attention_kernel_register_audit(tmp_path):
    """clip_attn257_kernel (the encoder's attention at S = 257) uses all 256 VGPRs by design: at most the ONE documented spill (an
    LDS offset for the token-256 pass, stored / reloaded once per item outside the key-block loop - DESIGN 4), none inside a block
    that holds MFMAs of the main pass."""
    import subprocess
    from tspo_amd import build as b
    asm = tmp_path / "clip_vit.s"
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           os.path.join(b.CSRC, "clip_vit.hip"), "-o", str(asm)])
    txt = open(asm).read()
    found = 0
    for m in re.finditer(r"\.name:\s+(\S*clip_attn257_kernel\S*)\n(.*?)\.wavefront_size", txt, flags=re.S):
        md = m.group(2)
        val = lambda key: int(re.search(key + r":\s+(\d+)", md).group(1))
        found += 1
        assert val(r"\.vgpr_spill_count") <= 1 and val(r"\.private_segment_fixed_size") <= 8 and val(r"\.sgpr_spill_count") == 0, \
            (m.group(1), val(r"\.vgpr_spill_count"), val(r"\.private_segment_fixed_size"))
    body = re.findall(r"^(_Z\S*clip_attn257_kernel\S*):[^\n]*\n(.*?)\n\.Lfunc_end", txt, flags=re.S | re.M)[0][1]
    scratch = [ln.strip().split()[0] for ln in body.split("\n") if ln.strip().startswith("scratch_")]
    assert sorted(scratch) in ([], ["scratch_load_dword", "scratch_store_dword"]), scratch      # one store + one reload per item, or none


def test_build_names_the_validated_toolchain():
    """the wait states around the inline-asm MFMAs were validated with ONE compiler; building with another, or with
    extra flags, must say so and name the post-build checks (it is not an error)."""
    from tspo_amd import build as b
    assert b.toolchain_note(b._hipcc(), []) == "", "this image's hipcc is the validated toolchain"
    note = b.toolchain_note(b._hipcc(), ["-O2"])
    assert "TSPO_EXTRA_HIPCC_FLAGS=-O2" in note and "code_audit" in note
    assert "not the validated" in b.toolchain_note("/bin/echo", [])
    ident = b.lib_identity()
    assert ident["mode"] == "product" and len(ident["src_sha256"]) == 64 and len(ident["lib_sha256"]) == 64


def test_option_flags_of_the_header_match_the_binding():
    """The option words of the `_ex` entries are plain #defines in include/tspo_hip.h; the ctypes side (tspo_amd/ops.py, _lib.py)
    restates them - they must agree (round 6 added TSPO_SEL_BF16)."""
    import re
    from tspo_amd import _lib, ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "tspo_hip.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(TSPO_[A-Z0-9_]+)\s+(\d+|0x[0-9a-fA-F]+)\s*$", hdr, flags=re.M)}
    assert defs["TSPO_SEL_BF16X3"] == ops.SEL_BF16X3 and defs["TSPO_SEL_ACCUMULATE"] == ops.SEL_ACCUMULATE and defs["TSPO_SEL_BF16"] == ops.SEL_BF16
    assert len({ops.SEL_BF16X3, ops.SEL_ACCUMULATE, ops.SEL_BF16}) == 3 and (ops.SEL_BF16X3 | ops.SEL_ACCUMULATE | ops.SEL_BF16) == 7
    for name in ("TSPO_CLIP_NO_LN_FOLD", "TSPO_CLIP_PRUNE_LAST", "TSPO_CLIP_FOLD_CACHED"):
        assert defs[name] == getattr(_lib, name), name
    assert ops._sel_flags("bf16", forward=True) == defs["TSPO_SEL_BF16"]
    import pytest as _pt
    with _pt.raises(ValueError):
        ops._sel_flags("bf16")                      # the backward calls do not take it
    with _pt.raises(ValueError):
        ops._sel_flags("bf16", accumulate=True)
