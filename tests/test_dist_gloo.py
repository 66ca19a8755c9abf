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


def test_shard_prompts_covers_everything():
    from tspo_amd.dist import shard_prompts
    for n in (1, 4, 7, 32):
        for w in (1, 2, 4, 8):
            got = [i for r in range(w) for i in shard_prompts(n, w, r)]
            assert got == list(range(n))
