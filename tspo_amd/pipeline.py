"""Device-resident pipelines over the HIP ops: frame scoring (inference) and the
TSPO policy step (training).  These are the two things bench.py times.

FrameScorer  : pixels -> CLIP-L features -> clip score -> selector -> greedy top-k
               (TSPOModel.forward without the CPU PIL/tokeniser stages;
               model/temporal_agent.py:177-185).
PolicyTrainer: the policy side of LLaVAVideoTSPOTrainer.compute_loss
               (tspo_trainer.py:496-609) with the redundancy removed: the selector
               runs ONCE per prompt (the reference runs it 2*G times with identical
               results), all G Gumbel-top-k rollouts are one launch, the policy
               gradient w.r.t. the logits is closed-form, one selector backward,
               one all-reduce of the flat gradient bucket, fused AdamW.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import ops


class FrameScorer:
    def __init__(self, clip_weights: ops.ClipVitWeights, selector_flat: torch.Tensor, dim: int = 768, heads: int = 8,
                 window_size: int = 12, score_tau: float = 0.025, fold_layernorm: bool = True,
                 prune_last_layer: bool = False, selector_precision: str = "fp32"):
        # selector_precision: "fp32" (default; greedy indices identical to the fp32 oracle) or "bf16" - the reference's own inference
        # precision (gen_id_tspo.py:55): bf16 GEMM operands, fp32 accumulation (ops.selector_forward(precision="bf16"))
        ops._sel_flags(selector_precision, forward=True)
        self.selector_precision = selector_precision
        self.clip, self.flat = clip_weights, selector_flat
        self.dim, self.heads, self.window, self.tau = dim, heads, window_size, score_tau
        self.fold_layernorm, self.prune_last_layer = fold_layernorm, prune_last_layer   # encoder options (ops.clip_vit_forward)
        self._sel_ws = None

    def _encode(self, px: torch.Tensor) -> torch.Tensor:
        return ops.clip_vit_forward(self.clip, px, fold_layernorm=self.fold_layernorm, prune_last_layer=self.prune_last_layer)

    def encode(self, pixels: torch.Tensor, shard_frames: bool = False, group=None) -> torch.Tensor:
        """pixels [B,T,3,H,W] or [N,3,H,W] -> features f32 [.., proj].  shard_frames=True splits the frames of the
        batch across the ranks of `group` and all-gathers the features (long-video option, tspo_amd.dist.sharded_apply)."""
        if shard_frames:
            from .dist import sharded_apply
            flat = pixels.reshape(-1, *pixels.shape[-3:])
            feats = sharded_apply(self._encode, flat, group)
            return feats.view(*pixels.shape[:-3], -1)
        if pixels.ndim == 5:
            B, T = pixels.shape[:2]
            return self._encode(pixels.reshape(B * T, *pixels.shape[2:])).view(B, T, -1)
        return self._encode(pixels)

    def score(self, feats: torch.Tensor, text_features: torch.Tensor, clip_scores: Optional[torch.Tensor] = None):
        """feats [B,T,D], text [B,M,D] -> scores f32 [B,T]."""
        if clip_scores is None:
            clip_scores = ops.clip_scores(text_features, feats)
        B, T, D = feats.shape
        need = ops._lib.lib().tspo_selector_workspace_bytes(B, T, D, self.heads, text_features.shape[1], self.window)
        if self._sel_ws is None or self._sel_ws.numel() < need:
            self._sel_ws = torch.empty((need,), dtype=torch.uint8, device=feats.device)
        scores, _, _ = ops.selector_forward(self.flat, feats, text_features, clip_scores, self.heads, self.window,
                                            self.tau, want_attn=False, ws=self._sel_ws, precision=self.selector_precision)
        return scores, clip_scores

    def __call__(self, pixels: torch.Tensor, text_features: torch.Tensor, k: int, method: str = "topk"):
        feats = self.encode(pixels)
        scores, clip = self.score(feats, text_features)
        idx = ops.topk_sorted(scores, k) if method == "topk" else ops.binmax(scores, k)
        return idx, scores, feats


class RolloutContext:
    """What `PolicyTrainer.rollout` saved for the matching backward: the selector workspace holding the activations
    of THAT forward, the shapes / temperature they belong to, and the parameter version they were computed with.
    Opaque to callers; `PolicyTrainer.backward` validates and consumes it (a context is good for one backward)."""
    __slots__ = ("ws", "shape", "M", "tau", "param_version", "serial", "consumed", "owner", "idx", "idx_version", "micro_steps")

    def __init__(self, ws, shape, M, tau, param_version, serial, owner, idx=None, micro_steps=1):
        self.ws, self.shape, self.M, self.tau = ws, tuple(shape), M, float(tau)
        self.param_version, self.serial, self.consumed, self.owner = param_version, serial, False, owner
        # the index tensor this rollout emitted (ascending by construction) - the TENSOR, which also pins its storage, and its
        # version counter: an address can be recycled by the allocator and `idx.copy_(...)` keeps it, identity + version cannot lie
        self.idx, self.idx_version = idx, (idx._version if idx is not None else -1)
        self.micro_steps = int(micro_steps)  # how many micro-steps of the accumulation window this batch holds


class PolicyTrainer:
    """Data-parallel TSPO policy step on one rank.  `flat`/`grad` are the fp32 buckets of ops.FLAT_LAYOUT.

    One optimizer step = `grad_accum_steps` micro-steps (train_deepspeed.sh:31 uses 2), each
        scores, idx, logp, ctx = rollout(feats, txt, clip, G, k, tau)      # selector forward once + G rollouts
        rewards = <frozen video-LLM pass, stock PyTorch>                    # [B, G]
        backward(ctx, feats, txt, logp, idx, rewards)                       # accumulates into the flat grad bucket
    and, on the accumulation boundary only, `optimizer_step()`: ONE all-reduce of the bucket
    (`reduce_fn`, default tspo_amd.dist.allreduce_bucket_ = RCCL over xGMI), clip-norm of the mean, fused AdamW.
    `step()` / `update()` wrap these for the common one-micro-batch case.

    Coalesced micro-steps (round 5).  The micro-steps of one optimizer step see the same weights (no update between them,
    tspo_trainer.py:500-552), so their policy math is one batch: `rollout(..., micro_steps=n)` takes the prompts of n micro-steps
    stacked along B ([n * B_micro, T, D]; equal T / k), draws every group's Gumbel noise from the Philox offset its own
    `rollout()` call would have used (indices bitwise those of the sequential path) and the matching `backward()` counts as n
    micro-steps - one forward, one sampler launch, one backward instead of n.  The reference's configuration
    (per_device_train_batch_size 1 x gradient_accumulation_steps 2, train_deepspeed.sh:30-31): 15 launches per optimizer step
    instead of 28.  `tspo_amd.train` does this whenever the micro-batches of a window stack.
    """

    def __init__(self, flat: torch.Tensor, dim: int = 768, heads: int = 8, window_size: int = 12,
                 lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: float = 1.0, seed: int = 2024, process_group=None, gemm_precision: str = "fp32",
                 grad_accum_steps: int = 1, lr_schedule: Optional[Callable[[int], float]] = None,
                 reduce_fn: Optional[Callable] = None, rank: Optional[int] = None):
        # gemm_precision: "fp32" (exact fp32 MFMA, default) or "bf16x3" (opt-in split-precision GEMMs, ~1e-5 relative
        # error; the reference trains in bf16, train_deepspeed.sh:33)
        ops._sel_flags(gemm_precision)
        self.gemm_precision = gemm_precision
        self.flat = flat
        self.grad = torch.zeros_like(flat)
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.dim, self.heads, self.window = dim, heads, window_size
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.lr_schedule = lr_schedule          # optimizer step (0-based) -> learning rate; None = constant `lr`
        self.n_train = ops.trainable_numel(dim)
        self.seed, self.step_no = seed, 0       # step_no = optimizer steps taken (AdamW bias correction)
        self.pg = process_group
        self._default_reduce = reduce_fn is None
        if reduce_fn is None:
            from .dist import allreduce_bucket_
            reduce_fn = allreduce_bucket_
        self.reduce_fn = reduce_fn              # (bucket, n, group) -> world size; leaves the SUM over ranks in bucket[:n]
        self._reduce_default = reduce_fn
        self.grad_accum_steps = int(grad_accum_steps)
        assert self.grad_accum_steps >= 1
        self._micro = 0                         # micro-steps accumulated since the last optimizer step
        self._rollouts = 0                      # rollout() calls so far: Philox offset (fresh noise for every call)
        self._param_version = 0                 # bumped by optimizer_step(): a context from before it is stale
        self._ws_pool = []                      # selector workspaces returned by consumed contexts
        self._norm_out = torch.empty((2,), dtype=torch.float32, device=flat.device)
        self._norm_ws = torch.empty((2048,), dtype=torch.uint8, device=flat.device)
        self._norm_part = torch.empty((2048,), dtype=torch.float32, device=flat.device)
        self._norm_np = 0                       # > 0: the last backward left the bucket's partial sums of squares there
        self._norm_grad_version = -1            # ... for the bucket contents of THIS tensor version (an in-place edit invalidates them)
        self._rank = rank

    # ---- topology -----------------------------------------------------------------------------------------
    def world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    def rank(self) -> int:
        if self._rank is not None:
            return self._rank
        import torch.distributed as dist
        return dist.get_rank(self.pg) if (dist.is_available() and dist.is_initialized()) else 0

    @staticmethod
    def _dist_initialised() -> bool:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    def _rank_seed(self) -> int:
        """Philox key of this rank: every data-parallel rank must draw DIFFERENT Gumbel noise for its prompts (the
        kernel's counter is the local (b, g, t)); golden-ratio stride in the 64-bit key space, rank 0 keeps `seed`."""
        return (self.seed + 0x9E3779B97F4A7C15 * self.rank()) & (2 ** 64 - 1)

    # ---- micro-step ---------------------------------------------------------------------------------------
    def rollout(self, feats, txt, clip, G: int, k: int, tau: float, noise=None, micro_steps: int = 1):
        """scores once, G Gumbel-top-k rollouts per prompt in one launch (tspo_trainer.py:508-537).
        Returns (scores [B,T], idx [B,G,k], logp [B,T], ctx) - pass `ctx` to the matching backward().
        micro_steps = n > 1: the batch stacks the prompts of n micro-steps of this accumulation window (class docstring)."""
        B, T, D = feats.shape
        n = int(micro_steps)
        if n < 1 or B % n or (n > 1 and self._micro + n > self.grad_accum_steps):
            raise ValueError(f"rollout(micro_steps={n}): B={B} must be a multiple of it and {self._micro} + {n} micro-steps must fit "
                             f"the accumulation window of {self.grad_accum_steps}")
        need = ops._lib.lib().tspo_selector_workspace_bytes(B, T, D, self.heads, txt.shape[1], self.window)
        ws = None
        for i, cand in enumerate(self._ws_pool):
            if cand.numel() >= need and cand.device == feats.device:
                ws = self._ws_pool.pop(i)
                break
        if ws is None:
            ws = torch.empty((need,), dtype=torch.uint8, device=feats.device)
        scores, _, _ = ops.selector_forward(self.flat, feats, txt, clip, self.heads, self.window, tau, want_attn=False,
                                            ws=ws, precision=self.gemm_precision)
        out = ops.gumbel_topk(scores, k, G, noise=noise, seed=self._rank_seed(), offset=self._rollouts,
                              prompts_per_offset=B // n if n > 1 else 0)
        ctx = RolloutContext(ws, (B, T, D), txt.shape[1], tau, self._param_version, self._rollouts, self, out["idx"], n)
        self._rollouts += n
        return scores, out["idx"], out["logp"], ctx

    def backward(self, ctx: RolloutContext, feats, txt, logp, idx, rewards) -> Dict[str, torch.Tensor]:
        """advantage -> closed-form dL/dlogits -> selector backward, accumulated into the flat gradient bucket
        (mean over the micro-steps of this optimizer step, like DeepSpeed's gradient accumulation)."""
        if not isinstance(ctx, RolloutContext) or ctx.owner is not self:
            raise TypeError("backward() needs the RolloutContext returned by this trainer's rollout()")
        if ctx.consumed:
            raise RuntimeError("this RolloutContext was already used by a backward(); call rollout() again")
        if ctx.param_version != self._param_version:
            raise RuntimeError("stale RolloutContext: the parameters changed (optimizer_step) after its rollout()")
        if tuple(feats.shape) != ctx.shape or txt.shape[1] != ctx.M:
            raise ValueError(f"backward(): feats {tuple(feats.shape)} / txt M={txt.shape[1]} do not match the rollout "
                             f"this context belongs to ({ctx.shape}, M={ctx.M})")
        n = ctx.micro_steps
        if self._micro + n > self.grad_accum_steps:
            raise RuntimeError("gradient accumulation boundary reached: call optimizer_step() before another backward()")
        if not (idx is ctx.idx and idx._version == ctx.idx_version):
            # an index tensor that is not the one this rollout emitted (e.g. a reference-style `ts_ids` in selection order): the
            # gradient kernels find a frame's rollouts by binary search in ascending lists, so sort each rollout's list on
            # the device (membership - all the policy gradient uses - does not depend on the order; no host sync)
            idx = torch.sort(idx, dim=-1).values
        B = feats.shape[0]
        # the second.. micro-step ADDS its gradient into the bucket inside the backward's own kernels (TSPO_SEL_ACCUMULATE): no
        # scratch bucket, no add pass
        target, acc = self.grad, self._micro > 0
        scale = 1.0 / ((B // n) * self.grad_accum_steps)      # every micro-step weighs 1 / accum, its prompts 1 / B_micro
        self._norm_np = 0
        # single rank, the whole accumulation window in this one backward, default exchange: nothing touches the bucket between
        # this backward and AdamW, so the backward's last kernel (the split reduction that writes the bucket) also leaves its
        # sum of squares
        fuse_norm = (self.grad_accum_steps == n and self._default_reduce and self.reduce_fn is self._reduce_default
                     and self.world() == 1 and not self._dist_initialised())
        if idx.shape[1] <= 64 and fuse_norm:
            adv, loss, self._norm_np = ops.policy_backward(self.flat, target, feats, txt, rewards, logp, idx, self.heads,
                                                           self.window, ctx.tau, ctx.ws, scale=scale,
                                                           precision=self.gemm_precision, norm_partials=self._norm_part)
            self._norm_grad_version = self.grad._version
        elif idx.shape[1] <= 64:    # advantage -> dL/dscores inside the backward's first kernel (one launch less)
            adv, loss = ops.policy_backward(self.flat, target, feats, txt, rewards, logp, idx, self.heads, self.window, ctx.tau,
                                            ctx.ws, scale=scale, precision=self.gemm_precision, accumulate=acc)
        else:
            adv, dlog, loss = ops.grpo_pg_grad(rewards, logp, idx, scale=scale)
            ops.selector_backward(self.flat, target, feats, txt, dlog, self.heads, self.window, ctx.tau, ctx.ws,
                                  precision=self.gemm_precision, accumulate=acc)
        self._micro += n
        ctx.consumed = True
        self._ws_pool.append(ctx.ws)
        ctx.ws, ctx.idx = None, None
        # `loss` is the kernels' UNSCALED per-prompt -sum(adv*...)/G (only dL/dscores carries `scale`): callers average it
        # over micro-steps themselves (train.py), so it is returned as is whatever grad_accum_steps is
        return {"loss": loss, "advantages": adv}

    def at_boundary(self) -> bool:
        return self._micro >= self.grad_accum_steps

    def current_lr(self) -> float:
        return float(self.lr_schedule(self.step_no)) if self.lr_schedule is not None else self.lr

    def optimizer_step(self, lr: Optional[float] = None) -> Dict[str, torch.Tensor]:
        """all-reduce (sum) of the bucket -> clip-norm of the MEAN gradient -> AdamW, on the accumulation boundary."""
        if self._micro == 0:
            raise RuntimeError("optimizer_step() without a backward()")
        world = int(self.reduce_fn(self.grad, self.n_train, self.pg))      # RCCL over xGMI (one collective)
        # the bucket now holds the SUM over ranks: its norm is world x the norm of the mean gradient, so clipping the
        # mean at max_norm == clipping the sum at world*max_norm, then scaling by 1/world (no extra pass over the bucket)
        use_lr = lr if lr is not None else self.current_lr()
        self.step_no += 1
        # (the partial sums describe the bucket as the backward left it: `grad` is a public tensor, so an in-place edit between
        # backward() and this call - a regulariser, a manual clip - falls back to the separate sum-of-squares pass)
        fused = (self._norm_np > 0 and world == 1 and self.reduce_fn is self._reduce_default
                 and self.grad._version == self._norm_grad_version)
        ns = ops.adamw_clip_step(self.flat, self.grad, self.m, self.v, self.n_train, use_lr, self.step_no, self.betas[0],
                                 self.betas[1], self.eps, self.wd, pre_scale=1.0 / world, max_norm=self.max_norm * world,
                                 out=self._norm_out, ws=self._norm_ws, norm_partials=self._norm_part if fused else None,
                                 n_partials=self._norm_np if fused else 0)
        self._norm_np = 0
        self._micro = 0
        self._param_version += 1
        return {"grad_norm_scale": ns, "lr": use_lr, "world": world}

    # ---- one-micro-batch conveniences ---------------------------------------------------------------------
    def update(self, ctx: RolloutContext, feats, txt, logp, idx, rewards, lr: Optional[float] = None):
        """backward() and, when that completes the accumulation window, optimizer_step()."""
        stats = self.backward(ctx, feats, txt, logp, idx, rewards)
        if self.at_boundary():
            stats.update(self.optimizer_step(lr))
        return stats

    def step(self, feats, txt, clip, reward_fn: Callable[[torch.Tensor], torch.Tensor], G: int, k: int, tau: float,
             noise=None, lr: Optional[float] = None, micro_steps: int = 1):
        scores, idx, logp, ctx = self.rollout(feats, txt, clip, G, k, tau, noise, micro_steps=micro_steps)
        rewards = reward_fn(idx)                      # [B,G] - the frozen video-LLM pass lives here (stock PyTorch)
        stats = self.update(ctx, feats, txt, logp, idx, rewards, lr)
        stats.update(scores=scores, idx=idx, rewards=rewards)
        return stats

    # ---- optimizer state (save_steps / resume; the reference saves model-only checkpoints, train_deepspeed.sh:41) ----
    def state_dict(self) -> Dict[str, object]:
        return {"flat": self.flat.detach().clone(), "exp_avg": self.m.clone(), "exp_avg_sq": self.v.clone(),
                "step": self.step_no, "rollouts": self._rollouts, "seed": self.seed, "micro": self._micro,
                "grad": self.grad.clone() if self._micro else None, "dim": self.dim}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        if int(sd["dim"]) != self.dim:
            raise ValueError(f"optimizer state is for dim={sd['dim']}, trainer has dim={self.dim}")
        self.flat.copy_(sd["flat"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        self.step_no, self._rollouts, self.seed = int(sd["step"]), int(sd["rollouts"]), int(sd["seed"])
        self._micro = int(sd.get("micro", 0))
        if self._micro and sd.get("grad") is not None:
            self.grad.copy_(sd["grad"])
        self._param_version += 1


def annealed_tau(score_tau: float, step: int, max_steps: int) -> float:
    """tspo_trainer.py:496: tau = tau0 - (tau0 - 0.01) / max_steps * step."""
    return score_tau - (score_tau - 0.01) / max_steps * step


def linear_decay_lr(base_lr: float, max_steps: int, warmup_steps: int = 0) -> Callable[[int], float]:
    """HF Trainer's default `lr_scheduler_type="linear"` (what train_deepspeed.sh runs with: lr 5e-4, no warmup):
    lr(step) = base * step/warmup during warmup, then base * (max_steps - step) / (max_steps - warmup)."""
    def f(step: int) -> float:
        if step < warmup_steps:
            return base_lr * step / max(1, warmup_steps)
        return base_lr * max(0.0, (max_steps - step) / max(1, max_steps - warmup_steps))
    return f
