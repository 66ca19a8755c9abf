"""GPU tests of the data-parallel TSPO step (BASELINE configs[3]) and of the long-video encode (configs[4]).

* two ranks on the ONE leased GPU (gloo process group with host staging - RCCL needs one device per rank), each
  running PolicyTrainer on its shard_prompts() share with the HIP kernels and exchanging the flat gradient bucket
  through tspo_amd.dist.allreduce_bucket_ (the same call the product issues over RCCL): the post-AdamW bucket must equal
  a single-process run on the concatenated batch, with a gradient large enough that the clip-norm-of-the-mean logic
  (sum over ranks, 1/world folded into the clip coefficient) actually clips;
* gradient accumulation (train_deepspeed.sh:31: 2 micro-steps, reduce on the boundary only) == one step on the
  concatenated micro-batches; misuse of a rollout context raises instead of silently using stale activations;
* a 4096-frame CLIP-L/14 encode: sampled frames against the oracle, the rest through size-independent properties;
  frame-sharded encode + all-gather on GPU tensors across the two ranks.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import tspo_oracle as O
from tspo_amd import ops, synth
from tspo_amd.pipeline import PolicyTrainer, linear_decay_lr

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

D, H, W, TAU = 64, 8, 12, 0.025
B_GLOBAL, T, G, K = 4, 96, 4, 8
MAX_NORM = 1e-3           # far below the raw gradient norm -> the clip coefficient is active
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(dev):
    sel = synth.selector_state(D, seed=3, std=0.1, bias_std=0.05)
    offs = ops.flat_offsets(D)
    flat = torch.zeros(offs["__total__"][0])
    for name, (off, shape) in offs.items():
        if not name.startswith("__"):
            flat[off:off + int(np.prod(shape))] = torch.from_numpy(sel[name]).flatten()
    return flat.to(dev)


def _batch(step):
    """Deterministic global batch of one optimizer step: features, text, clip scores, Gumbel noise, rewards."""
    f = torch.from_numpy(synth.normal((B_GLOBAL, T, D), 100 + step))
    t = torch.from_numpy(synth.normal((B_GLOBAL, 1, D), 200 + step))
    c = torch.stack([O.clip_cosine_scores(t[b], f[b]) for b in range(B_GLOBAL)])
    u = torch.from_numpy(synth.uniform((B_GLOBAL, G, T), 300 + step).reshape(B_GLOBAL, G, T)).clamp(1e-6, 1 - 1e-6)
    noise = -torch.log(-torch.log(u))
    rew = (torch.from_numpy(synth.uniform((B_GLOBAL, G), 400 + step)).round()
           + torch.from_numpy(synth.uniform((B_GLOBAL, G), 500 + step))).reshape(B_GLOBAL, G)
    return f.float(), t.float(), c.float(), noise.float(), rew.float()


def _run_steps(trainer, rows, dev):
    """STEPS optimizer steps on the prompts `rows` of every global batch; returns the first step's averaged grad too."""
    g0 = None
    for s in range(STEPS):
        f, t, c, noise, rew = (x[rows].to(dev) for x in _batch(s))
        st = trainer.step(f, t, c, lambda idx: rew, G, K, TAU, noise=noise)
        if s == 0:
            g0 = (trainer.grad[: trainer.n_train] * float(st["grad_norm_scale"][1])).cpu()   # clipped mean gradient
            assert float(st["grad_norm_scale"][1]) * st["world"] < 0.5, "the test must exercise the clip path"
    return g0


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from tspo_amd import dist as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    td.init_from_env("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    tr = PolicyTrainer(_flat(dev), dim=D, heads=H, window_size=W, lr=5e-4, max_grad_norm=MAX_NORM)
    assert tr.world() == world and tr.rank() == rank
    rows = list(td.shard_prompts(B_GLOBAL, world, rank))
    g0 = _run_steps(tr, rows, dev)
    # frame-sharded apply + all-gather on GPU tensors (long-video option): 7 rows over 2 ranks, uneven
    x = torch.arange(7 * 3, dtype=torch.float32, device=dev).view(7, 3)
    y = td.sharded_apply(lambda t_: t_ * 2 + 1, x)
    assert y.is_cuda and torch.equal(y, x * 2 + 1)
    # in-kernel Philox noise must differ between ranks (same local prompt index, same seed)
    f, t, c, _, _ = (z[rows].to(dev) for z in _batch(0))
    _, idx, _, ctx = tr.rollout(f, t, c, G, K, TAU)
    q.put((rank, rows, tr.flat.cpu().numpy(), g0.numpy(), idx.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_two_ranks_one_gpu_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t_: t_[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1] and res[1][1] == [2, 3]
    # single-process reference on the concatenated batch (no process group in this process -> world 1)
    tr = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, lr=5e-4, max_grad_norm=MAX_NORM)
    assert tr.world() == 1
    g_ref = _run_steps(tr, list(range(B_GLOBAL)), DEV).numpy()
    want = tr.flat.cpu().numpy()
    n = tr.n_train
    for rank, _, flat, g0, _ in res:
        # the clipped mean gradient of step 0 (sum over ranks x coefficient/world) == single-process clipped gradient
        np.testing.assert_allclose(g0, g_ref, rtol=1e-4, atol=1e-6 * np.abs(g_ref).max())
        # post-AdamW parameters after 3 steps
        np.testing.assert_allclose(flat[:n], want[:n], rtol=1e-5, atol=5e-4 * 1e-3)
        np.testing.assert_array_equal(flat[n:], want[n:])                 # ffn_o never touched
    np.testing.assert_array_equal(res[0][2], res[1][2])                    # replicas stay bit-identical
    assert not np.array_equal(res[0][4], res[1][4]), "ranks drew identical Gumbel noise"


def test_gradient_accumulation_equals_concatenated_batch():
    """2 micro-steps of 2 prompts each, reduce + AdamW on the boundary only == one step on the 4 prompts."""
    f, t, c, noise, rew = (x.to(DEV) for x in _batch(0))
    one = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, max_grad_norm=MAX_NORM)
    one.step(f, t, c, lambda idx: rew, G, K, TAU, noise=noise)
    acc = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, max_grad_norm=MAX_NORM, grad_accum_steps=2)
    calls = []
    acc.reduce_fn = lambda bucket, n, group: calls.append(n) or 1          # injectable exchange: count the collectives
    s1 = acc.step(f[:2], t[:2], c[:2], lambda idx: rew[:2], G, K, TAU, noise=noise[:2])
    assert "grad_norm_scale" not in s1 and calls == [] and acc.step_no == 0          # no reduce / update off the boundary
    assert torch.equal(acc.flat, _flat(DEV))
    s2 = acc.step(f[2:], t[2:], c[2:], lambda idx: rew[2:], G, K, TAU, noise=noise[2:])
    assert "grad_norm_scale" in s2 and calls == [acc.n_train] and acc.step_no == 1
    n = one.n_train
    np.testing.assert_allclose(acc.flat[:n].cpu().numpy(), one.flat[:n].cpu().numpy(), rtol=1e-5, atol=5e-4 * 1e-3)
    # the per-prompt loss a micro-step reports does not depend on the accumulation setting (it is averaged by the caller)
    ref_loss = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, max_grad_norm=MAX_NORM).step(
        f, t, c, lambda idx: rew, G, K, TAU, noise=noise)["loss"]
    assert torch.equal(torch.cat([s1["loss"], s2["loss"]]), ref_loss)
    # linear-decay schedule of the reference's HF trainer: lr(step) drives the fused AdamW
    sch = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, lr_schedule=linear_decay_lr(5e-4, 10))
    st = sch.step(f, t, c, lambda idx: rew, G, K, TAU, noise=noise)
    assert abs(st["lr"] - 5e-4) < 1e-12 and abs(sch.current_lr() - 4.5e-4) < 1e-12


@pytest.mark.parametrize("dim,heads,Tn,Gn,kn,per", [(768, 8, 512, 8, 16, 1), (D, H, 96, 4, 8, 2)], ids=["reference-1x2-D768-T512", "2x2-D64-T96"])
def test_coalesced_micro_steps_equal_the_sequential_accumulation(dim, heads, Tn, Gn, kn, per):
    """The reference trains with per_device_train_batch_size 1 x gradient_accumulation_steps 2 (train_deepspeed.sh:30-31) and does
    not update the weights between the micro-steps (tspo_trainer.py:500-552): the window's prompts as ONE stacked rollout /
    backward (`micro_steps=2`) must reproduce the sequential path - scores and Gumbel-top-k indices (IN-KERNEL Philox noise: every
    prompt draws from the offset its own rollout() call would have used) bit for bit, log-probs and advantages bit for bit, the
    accumulated gradient bucket within 5e-6 of its largest element (summation order of the weight-gradient reductions), and the
    parameters after clip + AdamW; two optimizer steps, so the Philox offsets of the second window are checked too."""
    g = torch.Generator(device=DEV).manual_seed(31)
    if dim == D:
        flat0 = _flat(DEV)
    else:
        offs = ops.flat_offsets(dim)
        flat0 = torch.zeros(offs["__total__"][0], device=DEV)
        for name, (off, shape) in offs.items():
            if name.endswith(".weight"):
                flat0[off:off + dim * dim] = torch.randn(dim * dim, generator=g, device=DEV) * 0.02
    seq = PolicyTrainer(flat0.clone(), dim=dim, heads=heads, window_size=W, max_grad_norm=MAX_NORM, grad_accum_steps=2, seed=77)
    coa = PolicyTrainer(flat0.clone(), dim=dim, heads=heads, window_size=W, max_grad_norm=MAX_NORM, grad_accum_steps=2, seed=77)
    for step in range(2):
        f = torch.randn(2 * per, Tn, dim, generator=g, device=DEV)
        t = torch.randn(2 * per, 1, dim, generator=g, device=DEV)
        c = ops.clip_scores(t, f)
        rew = (torch.rand(2 * per, Gn, generator=g, device=DEV) > 0.5).float() + torch.rand(2 * per, Gn, generator=g, device=DEV)
        outs = []
        for m in range(2):
            sl = slice(m * per, (m + 1) * per)
            sc, idx, lp, ctx = seq.rollout(f[sl], t[sl], c[sl], Gn, kn, TAU)
            st = seq.backward(ctx, f[sl], t[sl], lp, idx, rew[sl])
            outs.append((sc, idx, lp, st["advantages"], st["loss"]))
        sc2, idx2, lp2, ctx2 = coa.rollout(f, t, c, Gn, kn, TAU, micro_steps=2)
        assert ctx2.micro_steps == 2 and coa._rollouts == seq._rollouts
        st2 = coa.backward(ctx2, f, t, lp2, idx2, rew)
        assert coa.at_boundary() and seq.at_boundary()
        for j, (a, b) in enumerate(zip((sc2, idx2, lp2, st2["advantages"], st2["loss"]), zip(*outs))):
            assert torch.equal(a, torch.cat(list(b))), ("scores", "indices", "logp", "advantages", "loss")[j]
        assert not torch.equal(idx2[0], idx2[per]), "the two micro-steps drew the same noise"
        n = seq.n_train
        gs, gc = seq.grad[:n], coa.grad[:n]
        assert (gs - gc).abs().max().item() <= 5e-6 * gs.abs().max().item(), (gs - gc).abs().max().item() / gs.abs().max().item()
        seq.optimizer_step()
        coa.optimizer_step()
        np.testing.assert_allclose(coa.flat[:n].cpu().numpy(), seq.flat[:n].cpu().numpy(), rtol=1e-5, atol=5e-4 * 1e-3)
        # the second window checks the Philox offsets bit for bit again: start it from identical parameters / moments (the
        # gradients above agree to rounding, not bitwise)
        coa.flat.copy_(seq.flat); coa.m.copy_(seq.m); coa.v.copy_(seq.v)
    # misuse: a window that does not fit, a batch that does not split
    with pytest.raises(ValueError):
        coa.rollout(f, t, c, Gn, kn, TAU, micro_steps=3)
    one = PolicyTrainer(flat0.clone(), dim=dim, heads=heads, window_size=W, grad_accum_steps=1)
    with pytest.raises(ValueError):
        one.rollout(f, t, c, Gn, kn, TAU, micro_steps=2)


def test_backward_sorts_foreign_or_edited_index_tensors():
    """the 'is this the tensor the rollout emitted' check is identity + version, not an address - an in-place edit of
    the emitted tensor (idx.copy_(reordered)) or a different tensor at a recycled address must still be sorted on the device."""
    f, t, c, noise, rew = (x.to(DEV) for x in _batch(0))
    ref = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, max_grad_norm=MAX_NORM)
    sc, idx, lp, ctx = ref.rollout(f, t, c, G, K, TAU, noise=noise)
    ref.backward(ctx, f, t, lp, idx, rew)
    for how in ("inplace", "recycled"):
        tr = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, max_grad_norm=MAX_NORM)
        sc, idx, lp, ctx = tr.rollout(f, t, c, G, K, TAU, noise=noise)
        rev = idx.flip(-1).clone()                         # descending lists: the same SETS, not ascending
        if how == "inplace":
            idx.copy_(rev)                                 # same tensor object, same address, new version
            use = idx
        else:
            ptr = idx.data_ptr()
            del idx
            ctx.idx = None                                 # drop the pin so the allocator may hand the block out again
            use = torch.empty_like(rev)
            use.copy_(rev)
            if use.data_ptr() != ptr:
                pass                                       # (allocator chose another block: the test still checks the foreign-tensor path)
        tr.backward(ctx, f, t, lp, use, rew)
        n = tr.n_train
        assert torch.equal(tr.grad[:n], ref.grad[:n]), how


def test_fused_gradient_norm_equals_the_separate_pass(monkeypatch):
    """Single rank, no accumulation: the backward's split reduction also leaves the bucket's sum of squares
    (tspo_policy_backward_ex -> tspo_adamw_clip_step_ex, one launch less, no second pass over the gradient).  Same
    gradient bit for bit, same norm / clip coefficient / parameters as the separate sum-of-squares pass to rounding
    (different summation order); any path on which the bucket can change in between keeps the separate pass."""
    f, t, c, noise, rew = (x.to(DEV) for x in _batch(0))
    seen = []
    real = ops.adamw_clip_step
    monkeypatch.setattr(ops, "adamw_clip_step", lambda *a, **k: (seen.append(k.get("norm_partials") is not None), real(*a, **k))[1])
    fused = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, max_grad_norm=MAX_NORM)
    sf = fused.step(f, t, c, lambda idx: rew, G, K, TAU, noise=noise)
    sep = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, max_grad_norm=MAX_NORM, reduce_fn=lambda bucket, n, group: 1)
    ss = sep.step(f, t, c, lambda idx: rew, G, K, TAU, noise=noise)
    acc = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W, max_grad_norm=MAX_NORM, grad_accum_steps=2)
    acc.step(f[:2], t[:2], c[:2], lambda idx: rew[:2], G, K, TAU, noise=noise[:2])
    acc.step(f[2:], t[2:], c[2:], lambda idx: rew[2:], G, K, TAU, noise=noise[2:])
    assert seen == [True, False, False]
    assert torch.equal(fused.grad, sep.grad)
    nf, ns = sf["grad_norm_scale"].cpu().numpy(), ss["grad_norm_scale"].cpu().numpy()
    np.testing.assert_allclose(nf, ns, rtol=2e-6)
    assert nf[1] < 0.5                                                      # the clip is active
    ref_norm = float(torch.linalg.vector_norm(sep.grad[: sep.n_train].double()))
    assert abs(nf[0] - ref_norm) <= 2e-6 * ref_norm
    n = fused.n_train
    np.testing.assert_allclose(fused.flat[:n].cpu().numpy(), sep.flat[:n].cpu().numpy(), rtol=1e-5, atol=1e-9)
    # a second step reuses nothing stale: partial count is consumed by optimizer_step
    assert fused._norm_np == 0
    fused.step(f, t, c, lambda idx: rew, G, K, TAU, noise=noise)
    assert seen[-1] is True


def test_policy_ops_validate_rollout_indices(monkeypatch):
    """The policy-gradient kernels find membership by binary search: index lists must be strictly ascending per rollout
    (checked under ops.DEBUG_CHECKS / TSPO_DEBUG_CHECKS=1), and a group needs at least two rollouts."""
    f, t, c, noise, rew = (x.to(DEV) for x in _batch(2))
    tr = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W)
    _, idx, logp, ctx = tr.rollout(f, t, c, G, K, TAU, noise=noise)
    with pytest.raises(ValueError, match="G >= 2"):
        ops.grpo_pg_grad(rew[:, :1], logp, idx[:, :1])
    monkeypatch.setattr(ops, "DEBUG_CHECKS", True)
    ops.grpo_pg_grad(rew, logp, idx)                                   # ascending lists pass
    with pytest.raises(ValueError, match="ascending"):
        ops.grpo_pg_grad(rew, logp, idx.flip(-1))
    # the trainer itself does not rely on the caller: an index tensor other than the one its rollout emitted (here: every list
    # reversed, like a reference-style `ts_ids` in selection order) is sorted on the device -> the same gradient, bit for bit
    monkeypatch.setattr(ops, "DEBUG_CHECKS", False)
    tr.backward(ctx, f, t, logp, idx.flip(-1), rew)
    ref = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W)
    _, idx2, logp2, ctx2 = ref.rollout(f, t, c, G, K, TAU, noise=noise)
    assert torch.equal(idx2, idx)
    ref.backward(ctx2, f, t, logp2, idx2, rew)
    assert torch.equal(tr.grad, ref.grad) and bool(tr.grad.abs().sum() > 0)


def test_rollout_context_is_validated():
    f, t, c, noise, rew = (x.to(DEV) for x in _batch(1))
    tr = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W)
    with pytest.raises(TypeError):
        tr.backward(None, f, t, None, None, rew)                                    # update before any rollout
    _, idxA, lpA, ctxA = tr.rollout(f[:2], t[:2], c[:2], G, K, TAU, noise=noise[:2])
    _, idxB, lpB, ctxB = tr.rollout(f[2:3], t[2:3], c[2:3], G, K, TAU, noise=noise[2:3])   # interleaved second rollout
    with pytest.raises(ValueError):
        tr.backward(ctxB, f[:2], t[:2], lpA, idxA, rew[:2])                         # B's activations with A's batch
    # A's context still holds A's activations although B ran in between: same gradient as an undisturbed run
    tr.backward(ctxA, f[:2], t[:2], lpA, idxA, rew[:2])
    ref = PolicyTrainer(_flat(DEV), dim=D, heads=H, window_size=W)
    _, i2, l2, c2 = ref.rollout(f[:2], t[:2], c[:2], G, K, TAU, noise=noise[:2])
    ref.backward(c2, f[:2], t[:2], l2, i2, rew[:2])
    assert torch.equal(tr.grad, ref.grad)
    with pytest.raises(RuntimeError):
        tr.backward(ctxA, f[:2], t[:2], lpA, idxA, rew[:2])                         # consumed
    tr.optimizer_step()
    with pytest.raises(RuntimeError):
        tr.backward(ctxB, f[2:3], t[2:3], lpB, idxB, rew[2:3])                      # stale: parameters changed
    # fresh noise on every rollout() call even without an optimizer step in between
    _, i3, _, _ = tr.rollout(f[:2], t[:2], c[:2], G, K, TAU)
    _, i4, _, _ = tr.rollout(f[:2], t[:2], c[:2], G, K, TAU)
    assert not torch.equal(i3, i4)
    # optimizer state round trip
    sd = tr.state_dict()
    tr2 = PolicyTrainer(torch.zeros_like(tr.flat), dim=D, heads=H, window_size=W)
    tr2.load_state_dict(sd)
    assert torch.equal(tr2.flat, tr.flat) and torch.equal(tr2.m, tr.m) and tr2.step_no == tr.step_no


def test_long_video_encode_T4096():
    """BASELINE configs[4] per-GPU share: 4096 CLIP-L/14 frames in ONE encode (M = 1 052 672 rows, ~23 GB workspace).
    8 sampled frames against the CPU oracle; everything else through properties that do not depend on the size:
    determinism, frame-order equivariance (bitwise), batch-size independence (bitwise vs a 64-frame encode)."""
    import bench
    torch.set_num_threads(min(32, torch.get_num_threads()))
    c = bench.CLIP_L14
    state = bench.random_clip_state(c, DEV)
    clipw = ops.ClipVitWeights(state, c, DEV)
    n = 4096
    gen = torch.Generator(device=DEV).manual_seed(77)
    px = torch.randint(0, 256, (n, 3, 224, 224), generator=gen, device=DEV, dtype=torch.uint8)
    feats = ops.clip_vit_forward(clipw, px).clone()
    assert feats.shape == (n, 768) and bool(torch.isfinite(feats).all())
    assert torch.equal(ops.clip_vit_forward(clipw, px), feats)                       # deterministic
    rev = torch.arange(n - 1, -1, -1, device=DEV)
    assert torch.equal(ops.clip_vit_forward(clipw, px[rev]), feats[rev])             # frames are independent, bitwise
    sub = torch.tensor([0, 1, 777, 1023, 1024, 2500, 4094, 4095], device=DEV)
    small = ops.clip_vit_forward(clipw, px[sub].repeat(8, 1, 1, 1))                   # 64-frame encode, same kernels
    assert torch.equal(small[:8], feats[sub])
    w = {k: (v.float().cpu().to(torch.bfloat16).float() if v.ndim >= 2 and "position_embedding" not in k else v.float().cpu())
         for k, v in state.items()}
    with torch.no_grad():
        ref = O.clip_vit_forward(w, O.clip_normalize_pixels(px[sub].cpu()), num_heads=c["heads"], patch=c["patch"])
    got = feats[sub].cpu()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=-1).min().item()
    print(f"\n[T=4096 encode] sampled frames vs oracle: max|err|/max|ref| {err:.4f}, min cos {cos:.6f}")
    assert err < 3e-2 and cos > 0.999


def _train_worker(rank, world, port, out_dir, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from tspo_amd import train as tt
    cfg = tt.TrainConfig(output_dir=out_dir, max_steps=3, num_generations=4, training_sample_len=8,
                         gradient_accumulation_steps=2, save_steps=2, dim=64, heads=8, seed=5,
                         per_device_train_batch_size=2)
    # weak relevance signal: the rollouts of a prompt get different rewards, so advantages and gradients are non-zero
    m = tt.train(cfg, tt.SyntheticFeatures(T=96, D=64, seed=5, device="cuda", signal=0.1), backend="gloo", resume=False)
    q.put((rank, m["flat"].cpu().numpy(), {k: v for k, v in m.items() if k != "flat"}))
    assert not dist.is_initialized()        # train() created the group, so train() left together and destroyed it


def test_training_driver_two_ranks_one_gpu(tmp_path):
    """`python -m tspo_amd.train` as launched by torch.distributed.run, two ranks on the one GPU (gloo): every rank draws
    its own shard of the data stream, the bucket is reduced on the accumulation boundary only, the replicas end bit-identical,
    rank 0 alone writes metrics and checkpoints, and the logged metrics are the cross-rank means."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out = str(tmp_path / "run")
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, out, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])                       # replicas bit-identical after 3 steps
    ma, mb = ({k: v for k, v in r[2].items() if k != "elapsed_s"} for r in res)
    assert ma["step"] == 3 and ma == mb                                         # same reduced metrics on both ranks
    import json
    lines = [json.loads(l) for l in open(os.path.join(out, "metrics.jsonl"))]
    assert all(np.isfinite(l["loss"]) and np.isfinite(l["grad_norm"]) for l in lines) and max(l["grad_norm"] for l in lines) > 0
    assert [l["step"] for l in lines] == [1, 2, 3]                              # written once (rank 0), not twice
    assert sorted(d for d in os.listdir(out) if d.startswith("checkpoint-")) == ["checkpoint-2", "checkpoint-3"]
