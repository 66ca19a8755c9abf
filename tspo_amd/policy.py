"""Training-side policy API of the reference (LlavaQwenForCausalLM.temporal_sampling /
inference_ts, llava/model/language_model/llava_qwen.py:131-176), HIP-backed.

``TemporalPolicy`` owns a ``MultiModal_Align`` (named ``multiModal_align`` like the reference
attribute, llava_qwen.py:67) and can be mixed into / attached to the frozen video-LLM, which stays on
stock PyTorch-ROCm.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .temporal_agent import MultiModal_Align, inference_ts
from .utils import gumbel_softmax


class TemporalPolicy(nn.Module):
    def __init__(self, dim: int = 768, num_heads: int = 8):
        super().__init__()
        self.multiModal_align = MultiModal_Align(dim, num_heads)

    def get_confidence_score(self):
        return self.multiModal_align

    def temporal_sampling(self, image_embeddings, text_features, clip_scores=None, sample_len=64, ts_ids=None,
                          window_size=None, score_tau=0.025, method=None, noise=None, seed=None):
        """llava_qwen.py:131-144.  method None -> rollout (ts_ids None: returns (idx.clone(), idx)) or
        re-evaluation (ts_ids given: returned unchanged; a fresh draw is still made, as in the reference);
        otherwise greedy inference.  Returns (sel_idx, logp_ts [T], confidence [T])."""
        confidence, _ = self.get_confidence_score()(image_embeddings, text_features, clip_scores, window_size, score_tau)
        assert confidence.ndim == 1
        if method is None:
            sel_idx, _probs, logp_ts = gumbel_softmax(confidence.unsqueeze(1), sample_len=sample_len, noise=noise,
                                                      seed=seed)
            sel_idx = (sel_idx.clone(), sel_idx) if ts_ids is None else ts_ids
            return sel_idx, logp_ts, confidence
        return self.inference_ts(confidence, sample_len=sample_len, method=method)

    def inference_ts(self, confidence, sample_len=64, method="topk"):
        """llava_qwen.py:146-176 (unknown method returns None there)."""
        if method not in ("aks", "topk", "bin-max"):
            print(f"sample_method: {method}")
            return None
        return inference_ts(confidence, method, sample_len)
