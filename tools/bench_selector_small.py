#!/usr/bin/env python
"""Selector forward / policy backward at the reference's micro-batch (B = 1, T = 512): time per call and a checksum
(tilings of the fp32 GEMMs keep every output element's contraction order, so the checksum must not move).
    python tools/bench_selector_small.py [B] [T]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tspo_amd import ops
from tspo_amd.pipeline import PolicyTrainer
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
G, k, tau = 8, 16, 0.025
gen = torch.Generator(device=dev).manual_seed(99)
feats = torch.randn(B, T, 768, generator=gen, device=dev)
txt = torch.randn(B, 1, 768, generator=gen, device=dev)
clip = ops.clip_scores(txt, feats)
rew = (torch.rand(B, G, generator=gen, device=dev) > 0.5).float() + torch.rand(B, G, generator=gen, device=dev)
flat = bench.flat_from_state(bench.random_selector_state(768, dev), 768, dev)
tr = PolicyTrainer(flat.clone())


def timeit(fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3


def fwd():
    return tr.rollout(feats, txt, clip, G, k, tau)


scores, idx, logp, ctx = fwd()
tr._ws_pool.append(ctx.ws)
h = hashlib.sha256(scores.cpu().numpy().tobytes()).hexdigest()[:12]


def fwd_only():
    s, i, l, c = fwd()
    tr._ws_pool.append(c.ws)


def full():
    tr.step(feats, txt, clip, lambda i: rew, G, k, tau)


t_f = timeit(fwd_only)
t_s = timeit(full)
hg = hashlib.sha256(tr.grad.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"B={B} T={T}: rollout (forward + sampler) {t_f:.1f} us, whole step {t_s:.1f} us  scores {h} grad {hg}")
