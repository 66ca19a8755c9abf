"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/tspo_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

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
    assert l.tspo_version() == 1


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
