#!/usr/bin/env python
"""Whole-encode check of library builds against each other (same convention as tools/ab_libs.sh: tmp_ab/lib_<name>.so): bitwise
repeatability of each build and agreement with the first one, LayerNorm folded and stand-alone.
    python tools/encode_variants.py frames name_a name_b ...        (run from the repository root on the GPU box)"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from tspo_amd import ops
    T, name, first = int(sys.argv[1]), sys.argv[3], sys.argv[4] == "1"
    DEV = torch.device("cuda", 0)
    c = bench.CLIP_L14
    clipw = ops.ClipVitWeights(bench.random_clip_state(c, DEV), c, DEV)
    px = torch.randint(0, 256, (T, 3, 224, 224), generator=torch.Generator(device=DEV).manual_seed(1234), device=DEV, dtype=torch.uint8)
    for fold in (False, True):
        f0 = ops.clip_vit_forward(clipw, px, fold_layernorm=fold).float()
        f1 = ops.clip_vit_forward(clipw, px, fold_layernorm=fold).float()
        ref_path = f"/tmp/enc_ref_{T}_{int(fold)}.pt"
        if first:
            torch.save(f0.cpu(), ref_path)
        ref = torch.load(ref_path).to(DEV)
        print(f"{name} fold={fold}: repeat maxdiff {(f0 - f1).abs().max().item():.4g}; vs the first build: "
              f"{(f0 - ref).abs().max().item() / ref.abs().max().item():.4g} of range", flush=True)
else:
    T, names = sys.argv[1], sys.argv[2:]
    for i, n in enumerate(names):
        shutil.copy(os.path.join(ROOT, "tmp_ab", f"lib_{n}.so"), os.path.join(ROOT, "tspo_amd", "libtspo_hip.so"))
        subprocess.call([sys.executable, __file__, T, "--child", n, "1" if i == 0 else "0"])
