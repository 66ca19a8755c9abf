"""Data-parallel plumbing of the TSPO step over torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).  Replaces the DeepSpeed ZeRO-3 launcher of the reference
(train_deepspeed.sh:14-16, scripts/zero3.json): only 2.95 M selector parameters train, so the whole
exchange is ONE all-reduce of a 11.8 MB fp32 bucket per optimizer step plus ONE tiny packed metrics
all-reduce (the reference issues 7 separate gathers, tspo_trainer.py:610-634)."""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def init_from_env(backend: str = "nccl", device: "torch.device | None" = None, force: bool = False) -> Tuple[int, int, int]:
    """(rank, world, local_rank); initialises the default group when WORLD_SIZE > 1 (or, with force=True, also for a
    single rank: a world-size-1 "nccl" group still loads librccl and binds the device, so the collective calls below
    really execute).  `device` binds the RCCL communicator to this rank's GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port()) if world == 1 else "29500"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        if world == 1:      # forced one-rank group: own store (under torch.distributed.run env:// would wait for the agent's)
            kw["store"] = dist.TCPStore("127.0.0.1", free_port(), 1, is_master=True)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def launched_by_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


_RENDEZVOUS_ERRORS = ("address already in use", "eaddrinuse", "connection refused", "failed to connect", "connect() timed out",
                      "the server socket has failed to listen", "the client socket has failed to connect")


def self_spawn(n: int, argv: Sequence[str], module: "str | None" = None, timeout: "float | None" = None) -> int:
    """`_self_spawn_once`, started again ONCE on a fresh port when the first attempt dies of a rendezvous error (the free port is
    found by binding and releasing it; another process can take it in between - rare, but a launcher should not fail a job on it).
    Anything else - a rank's own error, a timeout - is returned as it is."""
    import sys
    rc, tail = _self_spawn_once(n, argv, module, timeout)
    if rc not in (0, 124) and any(m in tail.lower() for m in _RENDEZVOUS_ERRORS):
        sys.stderr.write(f"self_spawn: rendezvous failed (exit code {rc}); starting the {n} ranks once more on a new port\n")
        rc, tail = _self_spawn_once(n, argv, module, timeout)
    return rc


def _self_spawn_once(n: int, argv: Sequence[str], module: "str | None" = None, timeout: "float | None" = None):
    """Start `n` ranks of this program on this node without torch.distributed.run (one process per GPU; what
    `train_deepspeed.sh:14-16`'s `torchrun --nproc_per_node` does): `python <argv[0]> <argv[1:]>` (or `python -m module
    <argv[1:]>`) n times with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, a free rendezvous port and
    the dmabuf IPC mode RCCL needs.  Children inherit stdout (rank 0 prints the result line); their stderr is forwarded live
    with a `[rank r]` prefix.  Returns the first non-zero exit code (the other ranks are terminated by PID as soon as one rank
    fails - also when the launcher itself is interrupted), else 0 - and the kept stderr lines of all ranks (for self_spawn's retry rule)."""
    import collections
    import subprocess
    import sys
    import threading
    import time
    port = free_port()
    cmd = [sys.executable] + (["-m", module] if module else [argv[0]]) + list(argv[1:])
    procs, tails, readers = [], [], []
    out_lock = threading.Lock()

    def forward(r, pipe, keep):
        # every rank's stderr is forwarded LIVE, line by line, with a [rank r] prefix (a hung RCCL rendezvous must show its
        # diagnostics while it hangs), and the last lines are kept for the launcher's own exit message
        for raw in iter(pipe.readline, b""):
            ln = raw.decode(errors="replace")
            keep.append(ln)
            with out_lock:
                sys.stderr.write(ln if n == 1 else f"[rank {r}] {ln}")
                sys.stderr.flush()
        pipe.close()

    rc, failed = 0, None
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TSPO_SELF_SPAWNED="1")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            p = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE)
            procs.append(p)
            tails.append(collections.deque(maxlen=40))
            t = threading.Thread(target=forward, args=(r, p.stderr, tails[-1]), daemon=True)
            t.start()
            readers.append(t)
        t0 = time.monotonic()
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    if code != 0 and rc == 0:
                        rc, failed = code, procs.index(p)
            if timeout is not None and time.monotonic() - t0 > timeout:
                rc = 124
            if live and rc == 0:
                time.sleep(0.05)
    finally:
        # a rank failed, timed out, or the launcher itself is being interrupted: stop exactly the processes started here
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        for p in live:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        for t in readers:
            t.join(timeout=5)
    if rc != 0:
        who = f"rank {failed}" if failed is not None else f"timeout after {timeout} s"
        tail = "".join(list(tails[failed])[-12:]) if failed is not None else ""
        sys.stderr.write(f"self_spawn: {who} of {n} exited with code {rc}; the other ranks were stopped.  Last lines of its stderr:\n{tail}")
        return rc, "".join("".join(t) for t in tails)
    return rc, ""


def shard_prompts(n_global: int, world: int, rank: int) -> range:
    """Contiguous shard of the global batch of prompts for `rank` (all G rollouts of a prompt stay on its
    rank, so the group statistics of the advantage never cross ranks - tspo_trainer.py:587-592)."""
    base, rem = divmod(n_global, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def _host_staged(group) -> bool:
    """gloo moves CUDA tensors through the host itself only in some builds; stage explicitly (tests: two ranks on ONE
    GPU cannot use RCCL, which needs one device per rank)."""
    return dist.get_backend(group) == "gloo"


def allreduce_bucket_(bucket: torch.Tensor, n: int, group=None, average: bool = False) -> int:
    """THE gradient exchange of the data-parallel step: one in-place all-reduce(SUM) of bucket[:n] over the ranks of
    `group` ("nccl" = RCCL over xGMI on MI355X).  Returns the world size; with average=True the result is divided by it
    (PolicyTrainer keeps the sum and folds 1/world into the clip coefficient instead of a second pass over the bucket).
    No-op (returns 1) when torch.distributed is not initialised.  bucket[:n] starts at the bucket's base address, so
    the view RCCL sees keeps the allocation's alignment."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world > 1 or dist.get_backend(group) == "nccl":      # a 1-rank RCCL group still executes the collective
        view = bucket[:n]
        if view.is_cuda and _host_staged(group):
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            view.copy_(host)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
        if average:
            view.div_(world)
    return world


def allreduce_mean_(bucket: torch.Tensor, n: int, group=None) -> torch.Tensor:
    """In-place mean over ranks of bucket[:n] (allreduce_bucket_ with average=True)."""
    allreduce_bucket_(bucket, n, group, average=True)
    return bucket


METRIC_KEYS = ("ts_length", "completion_length", "reward", "advantages", "reward_mean", "reward_std", "loss")


def pack_metrics(values: Dict[str, float], rewards_per_func: Sequence[float]) -> torch.Tensor:
    """One small vector for every metric the reference gathers (tspo_trainer.py:610-634)."""
    return torch.tensor([float(values.get(k, 0.0)) for k in METRIC_KEYS] + [float(x) for x in rewards_per_func] + [1.0],
                        dtype=torch.float64)


def reduce_metrics(packed: torch.Tensor, n_reward_funcs: int, reward_names: Sequence[str], group=None) -> Dict[str, float]:
    """Mean over ranks with ONE all-reduce (last slot counts ranks)."""
    t = packed.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = t.device
        if dist.get_backend(group) == "nccl" and not t.is_cuda:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t = t.to(dev)
    t = t / t[-1]
    out = {k: t[i].item() for i, k in enumerate(METRIC_KEYS)}
    for j, name in enumerate(reward_names[:n_reward_funcs]):
        out[f"rewards/{name}"] = t[len(METRIC_KEYS) + j].item()
    return out


def shard_rows(n: int, world: int, rank: int) -> range:
    """Contiguous shard of n rows (frames) for `rank`; shards differ by at most one row."""
    return shard_prompts(n, world, rank)


def sharded_apply(fn, x: torch.Tensor, group=None) -> torch.Tensor:
    """Long-video option (SURVEY C3 / BASELINE configs[4]): when there are fewer videos than GPUs, shard the FRAMES of
    one video across ranks for the encode and exchange the small result: every rank applies `fn` to its contiguous
    slice of x along dim 0 and the per-rank outputs [n_r, D] are all-gathered into the full [n, D] on every rank
    (one collective, 1.5 KB per frame).  fn must be row-wise independent (CLIP frame encode is)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return fn(x)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = x.shape[0]
    mine = shard_rows(n, world, rank)
    local = fn(x[mine.start:mine.stop]) if len(mine) else None
    per = (n + world - 1) // world                       # pad every shard to the largest one
    probe = local if local is not None else fn(x[:1])
    buf = torch.zeros((per,) + tuple(probe.shape[1:]), dtype=probe.dtype, device=probe.device)
    if local is not None:
        buf[: local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(probe.shape[1:]), dtype=probe.dtype, device=probe.device)
    if buf.is_cuda and _host_staged(group):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, buf.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, buf, group=group)
    parts = [out[r * per: r * per + len(shard_rows(n, world, r))] for r in range(world)]
    return torch.cat(parts, dim=0)
