"""Builds libtspo_hip.so (gfx950) in-tree with hipcc.  No torch dependency in
the library: plain HIP runtime + the C ABI of include/tspo_hip.h."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libtspo_hip.so")
SOURCES = ["sampler.hip", "selector.hip", "gemm_bf16.hip", "gemm_dma.hip", "clip_vit.hip", "preprocess.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_bf16.h"), os.path.join(CSRC, "gemm_epilogue.h"), os.path.join(CSRC, "gemm_agpr_common.h"),
           os.path.join(CSRC, "gemm_dma_kernel.h"), os.path.join(os.path.dirname(PKG), "include", "tspo_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


# The hand-placed wait states around inline-asm MFMAs (csrc/gemm_epilogue.h: `s_nop 3` in front, `s_nop 15; s_nop 7` behind) and the
# literal-AGPR kernels were validated with THIS compiler; hipcc's hazard recogniser does not see MFMAs inside asm, and how it
# materialises their operand tuples may change with the version or with extra flags.  A different toolchain is not an
# error, but it must pass the post-build checks before its library is trusted:
#   python -m pytest tests/test_abi.py -k "code_audit or dev_hooks"          (generated-code audit: no VALU write of an MFMA operand
#                                                                             within two instructions, no compiler AGPR use, spills)
#   python -m pytest tests -m gpu -k "remainder_phase_bitwise or residual_statistics or in_place"   (77-vs-83 bitwise, statistics, aliasing)
VALIDATED_TOOLCHAIN = "roc-7.2.0"      # `hipcc --version`: AMD clang 22.0.0git ... roc-7.2.0 (HIP 7.2.26015)


def toolchain_note(hipcc: str, flags) -> str:
    """'' when the library is being built with the validated compiler and no extra flags, else what to run afterwards."""
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception as e:      # pragma: no cover
        ver = f"(hipcc --version failed: {e})"
    why = []
    if VALIDATED_TOOLCHAIN not in ver:
        why.append(f"hipcc is not the validated {VALIDATED_TOOLCHAIN} toolchain ({(ver.splitlines() or ['?'])[0]})")
    if flags:
        why.append(f"TSPO_EXTRA_HIPCC_FLAGS={' '.join(flags)}")
    if not why:
        return ""
    return ("tspo_amd.build: " + "; ".join(why) + " - the wait states around the hand-written MFMAs were validated with the stock "
            "compiler and flags: run `pytest tests/test_abi.py -k code_audit` and the bitwise GPU tests (see tspo_amd/build.py) "
            "before trusting this library")


DEV_SOURCES = ["dev/gemm_dma_lab.hip", "dev/gemm_agpr.hip"]      # compiled into the library only by build(dev=True) / `--dev`
DEV_HEADERS = [os.path.join(CSRC, "dev", "gemm_dma_lab_kernel.h")]
MODE_STAMP = LIB + ".mode"      # "product" | "dev": which build the .so on disk is (a dev library must never pass for the shipped one)


def built_mode():
    try:
        return open(MODE_STAMP).read().strip()
    except OSError:
        return None


def needs_build(dev: bool = False) -> bool:
    if not os.path.exists(LIB) or built_mode() != ("dev" if dev else "product"):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    if dev:
        deps += [os.path.join(CSRC, s) for s in DEV_SOURCES] + DEV_HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, dev: bool = False, only=None) -> str:
    """The translation units compile in parallel (one hipcc per source).  dev=True links the laboratory units of csrc/dev/
    in as well (GEMM schedule A/Bs, a timing-only ablation, the K-step probe - what tools/probe_gemm_dma.py and the A/B
    variants of tools/bench_gemm.py need); the product sources themselves carry no conditional code and the shipped library
    is built WITHOUT them (tests/test_abi.py checks the binary).  only=[...] recompiles just those sources and relinks with
    the other objects as they are."""
    if not force and not needs_build(dev):
        return LIB
    hipcc = _hipcc()
    flags = os.environ.get("TSPO_EXTRA_HIPCC_FLAGS", "").split()
    note = toolchain_note(hipcc, flags)
    if note:
        print(note, file=sys.stderr, flush=True)
    sources = SOURCES + (DEV_SOURCES if dev else [])

    def compile_one(s):
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        if only is not None and s not in only and os.path.exists(o):
            return o
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + flags + \
              ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(MODE_STAMP, "w") as f:
        f.write("dev\n" if dev else "product\n")
    return LIB


def lib_identity() -> dict:
    """What a measurement of the library's kernels is bound to (profiles/*_gemm_hbm_traffic_*.json carry it, bench.py matches it):
    sha256 of the shared library on disk and sha256 over the product sources + headers it is built from (a box that rebuilds the
    library from the same sources runs the same kernels even where the linker's output differs by a byte)."""
    import hashlib

    def sha(paths):
        h = hashlib.sha256()
        for q in paths:
            h.update(os.path.basename(q).encode() + b"\0")
            with open(q, "rb") as f:
                h.update(f.read())
        return h.hexdigest()
    return {"lib_sha256": sha([LIB]) if os.path.exists(LIB) else None, "mode": built_mode(),
            "src_sha256": sha(sorted([os.path.join(CSRC, s) for s in SOURCES] + HEADERS))}


if __name__ == "__main__":
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    build(force=bool(only) or "--force" in sys.argv, dev="--dev" in sys.argv, only=only[0] if only else None)
    print(LIB)
