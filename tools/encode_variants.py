#!/usr/bin/env python
"""Whole-encode A/B of GEMM variants (--dev / --lab builds: TSPO_GEMM_VARIANT): repeatability and agreement with variant 82.
    python tools/encode_variants.py [frames]"""
import sys, os, subprocess
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if len(sys.argv) > 2:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, bench
    from tspo_amd import ops
    DEV = torch.device("cuda", 0)
    c = bench.CLIP_L14
    clipw = ops.ClipVitWeights(bench.random_clip_state(c, DEV), c, DEV)
    px = torch.randint(0, 256, (T, 3, 224, 224), generator=torch.Generator(device=DEV).manual_seed(1234), device=DEV, dtype=torch.uint8)
    for fold in (False, True):
        f0 = ops.clip_vit_forward(clipw, px, fold_layernorm=fold).float()
        f1 = ops.clip_vit_forward(clipw, px, fold_layernorm=fold).float()
        ref_path = f"/tmp/enc_ref_{T}_{int(fold)}.pt"
        if os.environ.get("TSPO_GEMM_VARIANT") == "82":
            torch.save(f0.cpu(), ref_path)
        ref = torch.load(ref_path).to(DEV)
        print(f"variant {os.environ.get('TSPO_GEMM_VARIANT')} fold={fold}: repeat maxdiff {(f0 - f1).abs().max().item():.4g}; vs variant 82: "
              f"{(f0 - ref).abs().max().item() / ref.abs().max().item():.4g} of range", flush=True)
else:
    for v in ("82", "77"):
        subprocess.call([sys.executable, __file__, str(T), "child"], env=dict(os.environ, TSPO_GEMM_VARIANT=v))
