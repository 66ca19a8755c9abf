"""TSPO policy training driver on MI355X - what `train_deepspeed.sh` + `src/open_tspo/tspo.py` +
`LLaVAVideoTSPOTrainer` do for the temporal agent, without DeepSpeed / TRL:

    python -m tspo_amd.train --gpus 8 --features /data/tspo_feats --output-dir ckpt/tspo --max-steps 1000
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        -m tspo_amd.train --features /data/tspo_feats --output-dir ckpt/tspo --max-steps 1000

(the first form spawns its own ranks, tspo_amd.dist.self_spawn; the second is what train_deepspeed.sh:14-16 does).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Only the 2.95 M selector parameters train,
so the whole exchange is ONE all-reduce of the flat fp32 gradient bucket per optimizer step (`PolicyTrainer`) plus one
packed metrics all-reduce; ZeRO-3 (scripts/zero3.json) has nothing to shard here.

Reference behaviour kept (file:line of the reference):
  * per_device_train_batch_size 1, gradient_accumulation_steps 2, lr 5e-4 with the HF Trainer's default linear decay,
    num_generations 8, window_size 12, training_sample_len 16, score_tau 0.025, save_steps 100, save_total_limit 8,
    save_only_model (train_deepspeed.sh:14-42)
  * temperature annealing tau(step) (tspo_trainer.py:496), sample_len halved for "general" items (:510-513)
  * rewards per function, 'specific' = sum / 'general' = accuracy + 1 (:554-573), group-relative advantage (:587-592)
  * metrics averaged over ranks and logged every step (:610-650) - here as JSON lines
  * checkpoints hold the selector under the training prefix `multiModal_align.` (scripts/merge_weights.py:19-24 reads it)
The frozen video-LLM that turns the selected frames into an answer is a plug-in (`reward_model`, stock PyTorch-ROCm);
without one the driver uses the temporal-localisation reward on needle-in-a-haystack masks plus a mask-overlap proxy for
the accuracy reward, which exercises the complete policy path.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import time
from dataclasses import asdict, dataclass
from typing import Callable, Dict, Iterator, List, Optional

import torch

from . import dist as tdist
from . import io as tio
from . import ops, rewards
from .pipeline import PolicyTrainer, annealed_tau, linear_decay_lr


@dataclass
class TrainConfig:
    output_dir: str = "ckpt/tspo"
    max_steps: int = 100
    num_generations: int = 8
    window_size: int = 12
    training_sample_len: int = 16
    score_tau: float = 0.025
    learning_rate: float = 5e-4
    gradient_accumulation_steps: int = 2
    per_device_train_batch_size: int = 1
    max_grad_norm: float = 1.0
    save_steps: int = 100
    save_total_limit: int = 8
    logging_steps: int = 1
    seed: int = 42
    dim: int = 768
    heads: int = 8
    gemm_precision: str = "fp32"
    coalesce_micro_steps: bool = True      # run the micro-steps of an optimizer step as one stacked batch when their shapes allow


class Batch:
    """One micro-batch of prompts on this rank: features [B,T,D], text [B,1,D], clip [B,T], mask [B,T] bool, type, and `meta`:
    ALWAYS a list of length B - entry b describes prompt b (e.g. {"path": ...} of its feature cache; None when the data set has
    nothing to say).  The same convention holds for a single micro-batch and for the stacked batch of a coalesced accumulation
    window (round 6, ADVICE r5: a reward plug-in that reads `batch.meta[b]` sees the same thing on stacked and unstacked steps)."""

    def __init__(self, feats, txt, clip, mask, item_type="specific", meta=None):
        B = feats.shape[0]
        if meta is None:
            meta = [None] * B
        elif not isinstance(meta, list):
            meta = [meta] * B if B == 1 else None
        if meta is None or len(meta) != B:
            raise ValueError(f"Batch.meta must be a list with one entry per prompt (B={B})")
        self.feats, self.txt, self.clip, self.mask, self.item_type, self.meta = feats, txt, clip, mask, item_type, meta


class SyntheticFeatures:
    """Seeded stand-in for the feature cache: unit-normal frame features in which the frames of a planted segment are
    pulled towards the text feature, so the clip score and the relevance mask carry signal."""

    def __init__(self, T=512, D=768, seed=0, device="cuda", signal=0.1):
        # how strongly the relevant frames point at the text: 0.1 -> clip-score margin of a few logits at tau 0.025, rollouts
        # differ in reward and gradients flow; 0.75 -> every rollout finds the whole segment, all advantages are zero
        self.signal = float(signal)
        self.T, self.D, self.seed, self.dev = T, D, seed, torch.device(device)

    def batches(self, rank: int, world: int, bs: int, skip: int = 0) -> Iterator[Batch]:
        """`skip` = micro-batches already consumed (resume): an index advance, nothing is materialised for them."""
        i = int(skip)
        while True:
            g = torch.Generator(device=self.dev).manual_seed(self.seed * 1_000_003 + (i * world + rank))
            f = torch.randn(bs, self.T, self.D, generator=g, device=self.dev)
            t = torch.randn(bs, 1, self.D, generator=g, device=self.dev)
            start = torch.randint(0, self.T - self.T // 8, (bs,), generator=g, device=self.dev)
            pos = torch.arange(self.T, device=self.dev)[None]
            mask = (pos >= start[:, None]) & (pos < start[:, None] + self.T // 8)
            f = f + self.signal * mask[..., None] * t
            yield Batch(f, t, ops.clip_scores(t, f), mask, "specific" if i % 4 else "general")
            i += 1


class FeatureCacheDataset:
    """Feature-cache files of the evaluation / training flow (tspo_amd.io: {"image","text","clip_scores","sampled_idx"},
    optional "mask" [T] bool and "type")."""

    def __init__(self, root: str, device="cuda"):
        self.files = sorted(glob.glob(os.path.join(root, "**", "*.pth"), recursive=True))
        if not self.files:
            raise FileNotFoundError(f"no *.pth feature caches under {root}")
        self.dev = torch.device(device)

    def batches(self, rank: int, world: int, bs: int, skip: int = 0) -> Iterator[Batch]:
        """`skip` = micro-batches already consumed (resume): the file cursor advances, no file is read for them."""
        assert bs == 1, "videos differ in length: one prompt per micro-batch, like the reference (per_device_train_batch_size 1)"
        mine = [self.files[i] for i in tdist.shard_prompts(len(self.files), world, rank)] or self.files
        pos = int(skip) % len(mine)
        while True:
            for path in mine[pos:]:
                stat = torch.load(path, map_location="cpu")
                img = stat["image"].to(self.dev).float()[None]
                txt = stat["text"].to(self.dev).float().reshape(1, -1, img.shape[-1])[:, :1]
                clip = stat["clip_scores"].to(self.dev).float().reshape(1, -1)
                mask = stat.get("mask", torch.ones(img.shape[1], dtype=torch.bool)).to(self.dev).reshape(1, -1)
                yield Batch(img, txt, clip, mask, stat.get("type", "specific"), {"path": path})
            pos = 0


def _stackable(window: List[Batch]) -> bool:
    a = window[0]
    return len(window) > 1 and all(b.feats.shape == a.feats.shape and b.txt.shape == a.txt.shape and b.mask.shape == a.mask.shape
                                   and b.item_type == a.item_type for b in window[1:])


def _stack(window: List[Batch]) -> Batch:
    return Batch(torch.cat([b.feats for b in window]), torch.cat([b.txt for b in window]), torch.cat([b.clip for b in window]),
                 torch.cat([b.mask for b in window]), window[0].item_type, [m for b in window for m in b.meta])


def mask_reward_model(idx: torch.Tensor, batch: Batch) -> torch.Tensor:
    """Default reward plug-in -> rewards_per_func [B, G, 2] = (accuracy proxy, temporal localisation).
    The plug-in contract (what `train(reward_model=...)` calls - the frozen video-LLM pass of tspo_trainer.py:554-573 lives behind
    it): `reward_model(idx [B,G,k] int64 ascending, batch) -> [B, G, F] float`, where `batch` holds B prompts - ONE micro-batch, or
    the stacked micro-batches of a coalesced accumulation window (then B = accum x per-device batch) - and `batch.meta[b]` is prompt
    b's meta in BOTH cases.  Rows of `idx`, `batch.feats`, `batch.mask` and `batch.meta` correspond.  The temporal
    column is the reference's reward (tspo.py:146-159); the accuracy column stands in for the frozen video-LLM's answer
    check: 1 when at least 40 % of the selected frames are relevant (the iou gate left commented at tspo.py:131-134)."""
    temporal = rewards.selection_mask_reward_gpu(idx, batch.mask)
    return torch.stack([(temporal > 0.4).float(), temporal], dim=-1)


REWARD_NAMES = ("accuracy_reward", "temporal_localization_reward")


def _ckpt_dirs(out: str) -> List[str]:
    ds = [d for d in glob.glob(os.path.join(out, "checkpoint-*")) if os.path.isdir(d)]
    return sorted(ds, key=lambda d: int(d.rsplit("-", 1)[1]))


def save_checkpoint(trainer: PolicyTrainer, cfg: TrainConfig, step: int) -> str:
    d = os.path.join(cfg.output_dir, f"checkpoint-{step}")
    os.makedirs(d, exist_ok=True)
    offs = ops.flat_offsets(trainer.dim)
    import math
    state = {k: trainer.flat[o:o + math.prod(s)].view(s).clone() for k, (o, s) in offs.items() if not k.startswith("__")}
    tio.save_selector_safetensors(state, os.path.join(d, "model.safetensors"), prefix=tio.TRAIN_PREFIX)
    torch.save(trainer.state_dict(), os.path.join(d, "optimizer.pt"))       # (save_only_model in the reference; kept for resume)
    with open(os.path.join(d, "trainer_state.json"), "w") as f:
        json.dump({"global_step": step, "config": asdict(cfg)}, f)
    for old in _ckpt_dirs(cfg.output_dir)[:-cfg.save_total_limit]:
        for p in glob.glob(os.path.join(old, "*")):
            os.remove(p)
        os.rmdir(old)
    return d


def train(cfg: TrainConfig, data, reward_model: Callable[[torch.Tensor, Batch], torch.Tensor] = mask_reward_model,
          flat: Optional[torch.Tensor] = None, backend: str = "nccl", resume: bool = True,
          log: Optional[Callable[[Dict], None]] = None, stop_after: Optional[int] = None) -> Dict:
    """Runs cfg.max_steps optimizer steps on this rank's shard (resuming from the newest checkpoint in cfg.output_dir);
    returns the last logged metrics.  `stop_after` ends the run early after that global step, with a checkpoint (tests)."""
    import torch.distributed as dist
    world_env, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local if torch.cuda.device_count() > local else 0)
    torch.cuda.set_device(dev)
    own_group = world_env > 1 and not dist.is_initialized()
    rank, world, local = tdist.init_from_env(backend, device=dev)
    if flat is None:
        g = torch.Generator(device=dev).manual_seed(cfg.seed)            # HF _init_weights: N(0, 0.02), zero bias
        offs = ops.flat_offsets(cfg.dim)
        flat = torch.zeros(offs["__total__"][0], dtype=torch.float32, device=dev)
        for k, (o, s) in offs.items():
            if k.endswith(".weight"):
                flat[o:o + cfg.dim * cfg.dim] = torch.randn(cfg.dim * cfg.dim, generator=g, device=dev) * 0.02
    trainer = PolicyTrainer(flat, dim=cfg.dim, heads=cfg.heads, window_size=cfg.window_size, lr=cfg.learning_rate,
                            max_grad_norm=cfg.max_grad_norm, seed=cfg.seed, grad_accum_steps=cfg.gradient_accumulation_steps,
                            lr_schedule=linear_decay_lr(cfg.learning_rate, cfg.max_steps), gemm_precision=cfg.gemm_precision)
    start = 0
    if resume and _ckpt_dirs(cfg.output_dir):
        last = _ckpt_dirs(cfg.output_dir)[-1]
        trainer.load_state_dict(torch.load(os.path.join(last, "optimizer.pt"), map_location=dev))
        start = int(json.load(open(os.path.join(last, "trainer_state.json")))["global_step"])
    os.makedirs(cfg.output_dir, exist_ok=True)
    logf = open(os.path.join(cfg.output_dir, "metrics.jsonl"), "a") if rank == 0 else None
    # deterministic data order across resumes: the stream starts at the first micro-batch not yet consumed
    it = data.batches(rank, world, cfg.per_device_train_batch_size, skip=start * cfg.gradient_accumulation_steps)
    last_metrics: Dict = {}
    t0 = time.perf_counter()
    for step in range(start, cfg.max_steps):
        tau = annealed_tau(cfg.score_tau, step, cfg.max_steps)
        window = [next(it) for _ in range(cfg.gradient_accumulation_steps)]
        # The micro-steps of one optimizer step see the same weights (tspo_trainer.py:500-552: no update between them), so when
        # their batches stack - same T, text rows, item type (hence k) - they run as ONE rollout / backward of the stacked batch
        # (PolicyTrainer, "coalesced micro-steps": same Philox draws per prompt as the sequential path, 15 launches instead of 28
        # for the reference's 1 x 2 configuration).  Videos of different length (feature caches) keep one micro-step each.
        groups = [window] if cfg.coalesce_micro_steps and _stackable(window) else [[b] for b in window]
        acc = torch.zeros(len(tdist.METRIC_KEYS) + len(REWARD_NAMES), dtype=torch.float32, device=dev)   # metrics stay on the device
        for grp in groups:
            b = grp[0] if len(grp) == 1 else _stack(grp)
            k = rewards.training_sample_len(cfg.training_sample_len, b.item_type)
            scores, idx, logp, ctx = trainer.rollout(b.feats, b.txt, b.clip, cfg.num_generations, k, tau, micro_steps=len(grp))
            rpf = reward_model(idx, b)                                           # [B, G, F]   (frozen video-LLM plug-in)
            B, G, F = rpf.shape
            rew = rewards.combine_rewards(rpf.reshape(B * G, F), b.item_type).reshape(B, G)
            st = trainer.backward(ctx, b.feats, b.txt, logp, idx, rew)
            w = len(grp) / cfg.gradient_accumulation_steps
            vals = {"ts_length": torch.tensor(float(k), device=dev), "reward": rew.mean(), "advantages": st["advantages"].mean(),
                    "reward_mean": rew.mean(dim=1).mean(), "reward_std": rew.std(dim=1).mean(), "loss": st["loss"].mean()}
            for j, key in enumerate(tdist.METRIC_KEYS):
                if key in vals:
                    acc[j] += w * vals[key]
            acc[len(tdist.METRIC_KEYS):] += w * rpf.reshape(B * G, F).mean(dim=0)
        ost = trainer.optimizer_step()
        if (step + 1) % cfg.logging_steps == 0:
            host = acc.tolist()                                                  # the step's ONE device -> host transfer
            acc_m = dict(zip(tdist.METRIC_KEYS, host))
            acc_r = host[len(tdist.METRIC_KEYS):]
            m = tdist.reduce_metrics(tdist.pack_metrics(acc_m, acc_r), len(REWARD_NAMES), REWARD_NAMES)
            m.update(step=step + 1, learning_rate=ost["lr"], score_tau=tau,
                     grad_norm=float(ost["grad_norm_scale"][0]) / ost["world"], elapsed_s=round(time.perf_counter() - t0, 3))
            last_metrics = m
            if logf:
                logf.write(json.dumps(m) + "\n")
                logf.flush()
            if log:
                log(m)
        stop = stop_after is not None and step + 1 >= stop_after
        if rank == 0 and ((step + 1) % cfg.save_steps == 0 or step + 1 == cfg.max_steps or stop):
            save_checkpoint(trainer, cfg, step + 1)
        if stop:
            break
    if logf:
        logf.close()
    if world > 1:
        # nobody leaves while rank 0 is still writing the last checkpoint (a launcher would tear the job down)
        torch.cuda.synchronize()
        dist.barrier()
        if own_group:
            dist.destroy_process_group()
    last_metrics["flat"] = trainer.flat
    return last_metrics


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--features", default=None, help="directory of feature-cache .pth files (default: synthetic features)")
    ap.add_argument("--frames", type=int, default=512, help="T of the synthetic features")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--gpus", type=int, default=0, help="ranks to run; without a launcher (no WORLD_SIZE) the driver spawns them itself")
    ap.add_argument("--no-resume", action="store_true")
    for name, val in asdict(TrainConfig()).items():
        if isinstance(val, bool):
            ap.add_argument("--" + name.replace("_", "-"), action=argparse.BooleanOptionalAction, default=val)
        else:
            ap.add_argument("--" + name.replace("_", "-"), type=type(val), default=val)
    a = ap.parse_args(argv)
    if a.gpus > 1 and not tdist.launched_by_torchrun():
        import sys
        return tdist.self_spawn(a.gpus, [""] + list(sys.argv[1:] if argv is None else argv), module="tspo_amd.train")
    cfg = TrainConfig(**{k: getattr(a, k) for k in asdict(TrainConfig())})
    data = FeatureCacheDataset(a.features) if a.features else SyntheticFeatures(T=a.frames, D=cfg.dim, seed=cfg.seed)
    m = train(cfg, data, backend=a.backend, resume=not a.no_resume,
              log=lambda r: print(json.dumps(r), flush=True) if int(os.environ.get("RANK", "0")) == 0 else None)
    return 0 if m else 1


if __name__ == "__main__":
    raise SystemExit(main())
