#!/usr/bin/env python
"""Policy-step micro-benchmark (configs[2]: B=4, T=512, G=8, k=16) for profiling.
    python tools/bench_policy.py [steps] [fp32|bf16x3] [dp|dpc]
`dp` = the reference's training configuration (train_deepspeed.sh:30-31): B = 1 per micro-step, 2 micro-steps per optimizer step,
the bucket all-reduce issued on a live one-rank nccl (= RCCL) group - bench.py's `rollouts_dp_path.sequential`; `dpc` = the same
window as ONE stacked rollout / backward (coalesced micro-steps, production since round 5: `rollouts_dp_path`)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tspo_amd import ops
from tspo_amd.pipeline import PolicyTrainer
dev = "cuda"
B, T, G, k, tau = 4, 512, 8, 16, 0.025
gen = torch.Generator(device=dev).manual_seed(99)
feats = torch.randn(B, T, 768, generator=gen, device=dev)
txt = torch.randn(B, 1, 768, generator=gen, device=dev)
clip = ops.clip_scores(txt, feats)
rew = (torch.rand(B, G, generator=gen, device=dev) > 0.5).float() + torch.rand(B, G, generator=gen, device=dev)
flat = bench.flat_from_state(bench.random_selector_state(768, dev), 768, dev)
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
DP = len(sys.argv) > 3 and sys.argv[3] in ("dp", "dpc")
COALESCE = len(sys.argv) > 3 and sys.argv[3] == "dpc"
if DP:
    pg = bench.one_rank_group("nccl", torch.device("cuda", 0))
    B, accum = 1, 2
    fl = [torch.randn(1, T, 768, generator=gen, device=dev) for _ in range(accum)]
    tl = [torch.randn(1, 1, 768, generator=gen, device=dev) for _ in range(accum)]
    cl = [ops.clip_scores(t, f) for t, f in zip(tl, fl)]
    rl = [(torch.rand(1, G, generator=gen, device=dev) > 0.5).float() + torch.rand(1, G, generator=gen, device=dev) for _ in range(accum)]
    tr = PolicyTrainer(flat, gemm_precision=prec, grad_accum_steps=accum)

    f2, t2, c2, r2 = torch.cat(fl), torch.cat(tl), torch.cat(cl), torch.cat(rl)

    def one():
        if COALESCE:
            tr.step(f2, t2, c2, lambda idx: r2, G, k, tau, micro_steps=accum)
            return
        for i in range(accum):
            tr.step(fl[i], tl[i], cl[i], lambda idx, i=i: rl[i], G, k, tau)
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"dp-path optimizer step ({prec}): {dt * 1e6:.1f} us -> {accum * G / dt:.0f} rollouts/s")
    sys.exit(0)
tr = PolicyTrainer(flat, gemm_precision=prec)
for _ in range(5):
    tr.step(feats, txt, clip, lambda i: rew, G, k, tau)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
t0 = time.perf_counter()
for _ in range(n):
    tr.step(feats, txt, clip, lambda i: rew, G, k, tau)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"policy step ({prec}): {dt * 1e3:.3f} ms -> {B * G / dt:.0f} rollouts/s")
