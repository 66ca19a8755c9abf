"""Reward glue of the TSPO step (host-side string / index arithmetic, as in the reference):
src/open_tspo/tspo.py:86-166 and tspo_trainer.py:554-573.  The text being scored comes from the frozen
video-LLM (stock PyTorch-ROCm, outside the HIP path)."""
from __future__ import annotations

import re
from typing import List, Sequence

import torch


def map_prediction_to_option(pred):
    """tspo.py:86-99: first standalone letter a-e of the lower-cased response, else False."""
    model_response = pred.strip().lower()
    matches = re.findall(r'(?<![a-z])[a-e](?![a-z])', model_response)
    if len(matches) < 1:
        return False
    return matches[0]


def accuracy_reward(completions, solution, sel_idxs=None, total_mask=None, **kwargs) -> List[float]:
    """tspo.py:101-143.  Symbolic verification (math_verify) is attempted when importable, then the
    option-letter match on the <answer>...</answer> body (or the whole solution)."""
    contents = [c[0]["content"] for c in completions]
    rewards = []
    for content, sol in zip(contents, solution):
        reward = 0.0
        try:
            from math_verify import parse, verify  # optional dependency of the reference
            if float(verify(parse(content), parse(sol))) > 0:
                reward = 1.0
        except Exception:
            pass
        if reward == 0.0:
            try:
                m = re.search(r"<answer>(.*?)</answer>", sol, re.DOTALL)
                ground_truth = m.group(1).strip() if m else sol.strip()
                if map_prediction_to_option(content) == map_prediction_to_option(ground_truth):
                    reward = 1.0
            except Exception:
                pass
        rewards.append(reward)
    return rewards


def temporal_localization_reward(completions, solution, sel_idxs, total_mask, **kwargs) -> List[float]:
    """tspo.py:146-159: fraction of the selected frames that fall inside the ground-truth mask."""
    rewards = []
    for sel_idx in sel_idxs:
        idx = sel_idx[1].detach().cpu()
        rewards.append(torch.sum(total_mask[idx]).item() / len(idx))
    return rewards


def format_reward(completions, **kwargs) -> List[float]:
    """tspo.py:161-166."""
    pattern = r"<think>.*?</think>\s*<answer>.*?</answer>"
    return [1.0 if re.match(pattern, c[0]["content"], re.DOTALL) else 0.0 for c in completions]


reward_funcs_registry = {"accuracy": accuracy_reward, "format": format_reward, "temporal": temporal_localization_reward}


def combine_rewards(rewards_per_func: torch.Tensor, item_type: str) -> torch.Tensor:
    """tspo_trainer.py:570-573: 'specific' items sum all reward functions; 'general' items use accuracy + 1."""
    if item_type == "specific":
        return rewards_per_func.sum(dim=1)
    return rewards_per_func[:, 0:1].sum(dim=1) + 1


def training_sample_len(base_len: int, item_type: str) -> int:
    """tspo_trainer.py:510-513: k is halved for 'general' items."""
    return base_len if item_type == "specific" else base_len // 2


def selection_mask_reward_gpu(idx: torch.Tensor, total_mask: torch.Tensor) -> torch.Tensor:
    """Batched temporal reward on device: idx [B,G,k] int64, total_mask [B,T] bool -> [B,G]."""
    B, G, k = idx.shape
    return torch.gather(total_mask.float().unsqueeze(1).expand(B, G, -1), 2, idx).sum(-1) / k
