"""World-size-2 data-parallel path on CPU (gloo): prompt sharding, ONE bucket all-reduce == single-process
sum over the micro-batches, packed metrics == per-metric means.  The same code runs with backend "nccl"
(RCCL over xGMI) on the GPUs; the HIP kernels themselves are covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from tspo_amd import dist as td, ops, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = td.init_from_env("gloo")
    assert (r, w) == (rank, world)
    D = 64
    n = ops.trainable_numel(D)
    total = ops.flat_offsets(D)["__total__"][0]
    shard = td.shard_prompts(5, world, rank)                         # global batch of 5 prompts (uneven)
    bucket = torch.zeros(total)
    for b in shard:                                                  # per-prompt "gradients"
        bucket[:n] += torch.from_numpy(synth.normal((n,), 1000 + b))
    bucket[n:] = 7.0                                                 # ffn_o region must not be touched
    summed = bucket.clone()
    assert td.allreduce_bucket_(summed, n) == world          # THE exchange PolicyTrainer.optimizer_step issues (sum kept)
    td.allreduce_mean_(bucket, n)                             # = allreduce_bucket_(average=True)
    assert torch.allclose(summed[:n] / world, bucket[:n], rtol=1e-6, atol=1e-7) and torch.equal(summed[n:], bucket[n:])
    metrics = td.reduce_metrics(td.pack_metrics(
        dict(ts_length=16, completion_length=10 + rank, reward=0.5 * rank, advantages=0.0, reward_mean=rank, reward_std=1.0),
        [1.0 * rank, 0.25]), 2, ["accuracy_reward", "temporal_localization_reward"])
    # frame-sharded apply + all-gather (uneven: 7 rows over 2 ranks) == applying fn to everything
    x = torch.arange(7 * 3, dtype=torch.float32).view(7, 3)
    y = td.sharded_apply(lambda t: t * 2 + 1, x)
    assert torch.equal(y, x * 2 + 1), y
    y1 = td.sharded_apply(lambda t: t.sum(dim=1, keepdim=True), x[:1])       # fewer rows than ranks
    assert torch.equal(y1, x[:1].sum(dim=1, keepdim=True))
    q.put((rank, list(shard), bucket.numpy(), metrics))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_bucket_allreduce_and_metrics():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from tspo_amd import ops, synth
    n = ops.trainable_numel(64)
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4]
    expect = sum(synth.normal((n,), 1000 + b).astype(np.float64) for b in range(5)) / world
    for _, _, bucket, m in res:
        np.testing.assert_allclose(bucket[:n], expect, rtol=1e-5, atol=1e-6)
        assert np.all(bucket[n:] == 7.0)
        assert abs(m["completion_length"] - 10.5) < 1e-9 and abs(m["reward"] - 0.25) < 1e-9
        assert abs(m["rewards/accuracy_reward"] - 0.5) < 1e-9 and abs(m["rewards/temporal_localization_reward"] - 0.25) < 1e-9
    np.testing.assert_array_equal(res[0][2], res[1][2])               # replicas stay bit-identical


def _worker8(rank, world, port, q):
    """configs[3] / configs[4] as they are partitioned on the 8 GPUs of a node, with the REAL sizes: 32 prompts over 8 ranks, the
    D = 768 bucket (2 952 960 trainable of 3 543 552 elements), a 4096-frame video sharded unevenly over the ranks."""
    sys.path.insert(0, ROOT)
    from tspo_amd import dist as td, ops
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    td.init_from_env("gloo")
    n, total = ops.trainable_numel(768), ops.flat_offsets(768)["__total__"][0]
    assert (n, total) == (2952960, 3543552)
    shard = td.shard_prompts(32, world, rank)
    bucket = torch.zeros(total)
    for b in shard:                                      # a cheap per-prompt "gradient": value depends on (prompt, element)
        bucket[:n] += (torch.arange(n, dtype=torch.float32) % 97 - 48.0) * (b + 1) * 1e-3
    bucket[n:] = -3.0
    assert td.allreduce_bucket_(bucket, n) == world
    # frame-sharded apply over 4096 + 5 frames (uneven shards) with a [*, 768] feature block per frame
    T = 4096 + 5
    x = (torch.arange(T, dtype=torch.float32)[:, None] * 0.5 + torch.arange(8, dtype=torch.float32)[None]).contiguous()
    y = td.sharded_apply(lambda t: t.repeat(1, 96) * 2.0, x)                   # [T, 768]
    ok = y.shape == (T, 768) and torch.equal(y, x.repeat(1, 96) * 2.0)
    m = td.reduce_metrics(td.pack_metrics(dict(ts_length=16, completion_length=0, reward=float(rank), advantages=0.0,
                                               reward_mean=float(rank), reward_std=1.0, loss=-0.5 * rank), [float(rank % 2), 0.25]),
                          2, ["accuracy_reward", "temporal_localization_reward"])
    q.put((rank, list(shard), float(bucket[:n].double().sum()), float(bucket[n:].sum()), bucket[:64].numpy(), ok, m))
    dist.barrier()
    dist.destroy_process_group()


def test_dp8_real_shard_sizes():
    """the N = 8 leg cannot run on hardware here, so its partitioning runs on CPU ranks with the real sizes: 32
    prompts -> 4 per rank, ONE all-reduce of the 2 952 960-element fp32 bucket (sum over all 32 prompts on every rank, ffn_o tail
    untouched, replicas bit-identical), a 4101-frame video sharded unevenly over 8 ranks and gathered back, packed metrics means."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [list(range(4 * i, 4 * i + 4)) for i in range(8)]
    n = 2952960
    base = (np.arange(n, dtype=np.float64) % 97 - 48.0)
    expect_sum = float((base * 1e-3).sum() * sum(b + 1 for b in range(32)))
    for _, _, tot, tail, head, ok, m in res:
        assert ok
        assert abs(tot - expect_sum) <= 1e-4 * abs(expect_sum) and tail == -3.0 * (3543552 - n)
        np.testing.assert_array_equal(head, res[0][4])                                  # replicas bit-identical
        assert abs(m["reward"] - 3.5) < 1e-6 and abs(m["loss"] + 1.75) < 1e-6 and abs(m["rewards/accuracy_reward"] - 0.5) < 1e-6
    np.testing.assert_allclose(res[0][4], (base[:64] * 1e-3 * sum(b + 1 for b in range(32))).astype(np.float32), rtol=1e-5)


def test_shard_prompts_covers_everything():
    from tspo_amd.dist import shard_prompts
    for n in (1, 4, 7, 32):
        for w in (1, 2, 4, 8):
            got = [i for r in range(w) for i in shard_prompts(n, w, r)]
            assert got == list(range(n))


def test_self_spawn_retries_once_on_a_rendezvous_error_only(tmp_path):
    """The launcher finds its rendezvous port by binding and releasing it; when another process takes the port in between, rank 0's
    store cannot listen ("address already in use").  self_spawn starts the ranks ONCE more on a fresh port for exactly that class
    of failure; a rank's own error is returned at once (no second attempt)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    marker = tmp_path / "first_attempt_done"
    prog = tmp_path / "prog.py"
    prog.write_text("import os, sys\n"
                    f"m = {str(marker)!r}\n"
                    "if os.environ['RANK'] == '0' and not os.path.exists(m):\n"
                    "    open(m, 'w').close()\n"
                    "    sys.stderr.write('RuntimeError: The server socket has failed to listen on any local network address. "
                    "port: 29511, useIpv6: false, code: -98, name: EADDRINUSE, message: address already in use\\n'); sys.exit(1)\n"
                    "sys.exit(0)\n")
    code = ("import sys; sys.path.insert(0, %r); from tspo_amd import dist as d; sys.exit(d.self_spawn(2, [%r], timeout=60))" % (root, str(prog)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "starting the 2 ranks once more on a new port" in r.stderr, r.stderr[-2000:]
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys\nsys.stderr.write('ValueError: my own bug\\n'); sys.exit(5 if os.environ['RANK'] == '1' else 0)\n")
    code2 = code.replace(str(prog), str(bad))
    r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=180)
    assert r2.returncode == 5 and "once more" not in r2.stderr
