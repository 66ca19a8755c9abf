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
                 window_size: int = 12, score_tau: float = 0.025):
        self.clip, self.flat = clip_weights, selector_flat
        self.dim, self.heads, self.window, self.tau = dim, heads, window_size, score_tau
        self._sel_ws = None

    def encode(self, pixels: torch.Tensor, shard_frames: bool = False, group=None) -> torch.Tensor:
        """pixels [B,T,3,H,W] or [N,3,H,W] -> features f32 [.., proj].  shard_frames=True splits the frames of the
        batch across the ranks of `group` and all-gathers the features (long-video option, tspo_amd.dist.sharded_apply)."""
        if shard_frames:
            from .dist import sharded_apply
            flat = pixels.reshape(-1, *pixels.shape[-3:])
            feats = sharded_apply(lambda px: ops.clip_vit_forward(self.clip, px), flat, group)
            return feats.view(*pixels.shape[:-3], -1)
        if pixels.ndim == 5:
            B, T = pixels.shape[:2]
            return ops.clip_vit_forward(self.clip, pixels.reshape(B * T, *pixels.shape[2:])).view(B, T, -1)
        return ops.clip_vit_forward(self.clip, pixels)

    def score(self, feats: torch.Tensor, text_features: torch.Tensor, clip_scores: Optional[torch.Tensor] = None):
        """feats [B,T,D], text [B,M,D] -> scores f32 [B,T]."""
        if clip_scores is None:
            clip_scores = ops.clip_scores(text_features, feats)
        B, T, D = feats.shape
        need = ops._lib.lib().tspo_selector_workspace_bytes(B, T, D, self.heads, text_features.shape[1], self.window)
        if self._sel_ws is None or self._sel_ws.numel() < need:
            self._sel_ws = torch.empty((need,), dtype=torch.uint8, device=feats.device)
        scores, _, _ = ops.selector_forward(self.flat, feats, text_features, clip_scores, self.heads, self.window,
                                            self.tau, want_attn=False, ws=self._sel_ws)
        return scores, clip_scores

    def __call__(self, pixels: torch.Tensor, text_features: torch.Tensor, k: int, method: str = "topk"):
        feats = self.encode(pixels)
        scores, clip = self.score(feats, text_features)
        idx = ops.topk_sorted(scores, k) if method == "topk" else ops.binmax(scores, k)
        return idx, scores, feats


class PolicyTrainer:
    """Data-parallel TSPO policy step on one rank.  `flat`/`grad` are the fp32 buckets of ops.FLAT_LAYOUT."""

    def __init__(self, flat: torch.Tensor, dim: int = 768, heads: int = 8, window_size: int = 12,
                 lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: float = 1.0, seed: int = 2024, process_group=None, gemm_precision: str = "fp32"):
        # gemm_precision: "fp32" (exact fp32 MFMA, default) or "bf16x3" (opt-in split-precision GEMMs, ~1e-5 relative
        # error; the reference trains in bf16, train_deepspeed.sh:33)
        ops._sel_flags(gemm_precision)
        self.gemm_precision = gemm_precision
        self.flat = flat
        self.grad = torch.zeros_like(flat)
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.dim, self.heads, self.window = dim, heads, window_size
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.n_train = ops.trainable_numel(dim)
        self.seed, self.step_no = seed, 0
        self.pg = process_group
        self._ws = None

    def world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    def rollout(self, feats, txt, clip, G: int, k: int, tau: float, noise=None):
        """scores once, G Gumbel-top-k rollouts per prompt in one launch (tspo_trainer.py:508-537)."""
        B, T, D = feats.shape
        need = ops._lib.lib().tspo_selector_workspace_bytes(B, T, D, self.heads, txt.shape[1], self.window)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=feats.device)
        scores, _, _ = ops.selector_forward(self.flat, feats, txt, clip, self.heads, self.window, tau, want_attn=False,
                                            ws=self._ws, precision=self.gemm_precision)
        out = ops.gumbel_topk(scores, k, G, noise=noise, seed=self.seed, offset=self.step_no)
        return scores, out["idx"], out["logp"]

    def update(self, feats, txt, logp, idx, rewards, tau: float, lr: Optional[float] = None) -> Dict[str, torch.Tensor]:
        """advantage -> closed-form dL/dlogits -> selector backward -> all-reduce(mean) -> clip -> AdamW."""
        B = feats.shape[0]
        adv = ops.grpo_advantage(rewards)
        dlog, loss = ops.pg_grad_logits(logp, idx, adv, scale=1.0 / B)
        ops.selector_backward(self.flat, self.grad, feats, txt, dlog, self.heads, self.window, tau, self._ws,
                              precision=self.gemm_precision)
        world = self.world()
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.grad[: self.n_train], op=dist.ReduceOp.SUM, group=self.pg)   # RCCL over xGMI
        ns = ops.grad_norm_scale(self.grad, self.n_train, 1.0 / world, self.max_norm * world)
        # ^ the bucket holds the SUM over ranks: its norm is world x the norm of the mean gradient, so clipping the
        #   mean at max_norm == clipping the sum at world*max_norm, then scaling by 1/world.
        self.step_no += 1
        ops.adamw_step(self.flat, self.grad, self.m, self.v, self.n_train, lr if lr is not None else self.lr,
                       self.step_no, self.betas[0], self.betas[1], self.eps, self.wd, 1.0, ns)
        return {"loss": loss, "advantages": adv, "grad_norm_scale": ns}

    def step(self, feats, txt, clip, reward_fn: Callable[[torch.Tensor], torch.Tensor], G: int, k: int, tau: float,
             noise=None, lr: Optional[float] = None):
        scores, idx, logp = self.rollout(feats, txt, clip, G, k, tau, noise)
        rewards = reward_fn(idx)                      # [B,G] - the frozen video-LLM pass lives here (stock PyTorch)
        stats = self.update(feats, txt, logp, idx, rewards, tau, lr)
        stats.update(scores=scores, idx=idx, rewards=rewards)
        return stats


def annealed_tau(score_tau: float, step: int, max_steps: int) -> float:
    """tspo_trainer.py:496: tau = tau0 - (tau0 - 0.01) / max_steps * step."""
    return score_tau - (score_tau - 0.01) / max_steps * step
