"""Drop-in for the reference's ``model/temporal_agent.py`` (same names, signatures,
state-dict keys and error behaviour), computing on MI355X through libtspo_hip.so.

    from tspo_amd.temporal_agent import TSPOModel, MultiModal_Align, positional_encoding

Reference: model/temporal_agent.py:10-19 (positional_encoding), :21-79
(Simple_SelfAttn), :81-143 (MultiModal_Align), :146-231 (TSPOModel).
There is no CPU / eager fallback: tensors must live on the GPU.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .utils import AKS_sampling, generate_uniform_integers  # noqa: F401  (re-exported like the reference module)


def positional_encoding(T, C):
    """[1,T,C] sinusoid table, position normalised by T (temporal_agent.py:10-19).
    Host-side utility with the reference's exact expression; the HIP selector
    evaluates the same formula in-kernel and never reads this table."""
    div_term = torch.exp(torch.arange(0, C, 2) * (-torch.log(torch.tensor(10000.0)) / C))
    pe = torch.zeros(1, T, C)
    position = torch.arange(T).unsqueeze(1) / T
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


class Simple_SelfAttn(nn.Module):
    """Parameter container with the reference's sub-module names (Self_q/Self_k/
    Self_v/ffn_o).  The arithmetic (q/k/v projections + banded softmax) runs
    fused inside ``MultiModal_Align.forward``; ``ffn_o`` exists in checkpoints
    but is never applied (temporal_agent.py:77-79)."""

    def __init__(self, dim=4096, num_heads=8, dropout=0.0) -> None:
        super().__init__()
        self.Self_q = nn.Linear(dim, dim)
        self.Self_k = nn.Linear(dim, dim)
        self.Self_v = nn.Linear(dim, dim)
        self.dropout = nn.Dropout(dropout)
        self.ffn_o = nn.Linear(dim, dim)
        self.embed_size = dim
        self.num_heads = num_heads
        self.head_dim = self.embed_size // num_heads


class _SelectorFn(torch.autograd.Function):
    """autograd bridge: forward/backward are the HIP selector kernels."""

    @staticmethod
    def forward(ctx, module, img, txt, clip, window, tau, *params):
        flat = module._flat_params(training=True)   # (grad mode is off inside Function.forward; this IS the training path)
        scores, attn, ws = ops.selector_forward(flat, img, txt, clip, module.num_heads, window, tau)
        ctx.module, ctx.window, ctx.tau, ctx.ws = module, window, tau, ws
        ctx.save_for_backward(img, txt)
        # the backward needs the weights THIS forward used.  Packed copy of non-flat parameters: it is repacked in place on
        # the next forward, so this graph keeps a private copy (14 MB).  Flat mode: the live bucket is used and its
        # version counter is recorded - an in-place update between forward and backward raises, as autograd does for
        # any saved tensor ("modified by an inplace operation").
        if module._is_flat():
            ctx.flat, ctx.flat_version = flat, flat._version
        else:
            ctx.flat, ctx.flat_version = flat.clone(), None
        ctx.mark_non_differentiable(attn)
        return scores, attn

    @staticmethod
    def backward(ctx, dscores, _dattn):
        img, txt = ctx.saved_tensors
        m = ctx.module
        if ctx.flat_version is not None and ctx.flat._version != ctx.flat_version:
            raise RuntimeError("MultiModal_Align: the flat parameter bucket was modified in place between this forward and "
                               "its backward (optimizer step / EMA swap with a graph still alive); run the forward again")
        fg = torch.zeros_like(ctx.flat)
        ops.selector_backward(ctx.flat, fg, img, txt, dscores.contiguous(), m.num_heads, ctx.window, ctx.tau, ctx.ws)
        offs = ops.flat_offsets(m.dim)
        grads = []
        for name, p in m._trainable_named():
            off, shape = offs[name]
            grads.append(fg[off:off + p.numel()].view(shape).to(p.dtype))
        return (None, None, None, None, None, None, *grads)


class MultiModal_Align(nn.Module):
    """Temporal scoring head (temporal_agent.py:81-143), HIP-backed.

    State-dict keys are the reference's (``temporal.Self_{q,k,v}.*``,
    ``temporal.ffn_o.*``, ``mlp.0.*``, ``mlp.2.*``).  ``flatten_parameters()``
    re-homes all parameters (and their grads) as views of one flat fp32 bucket,
    which is what the data-parallel trainer all-reduces.
    """

    def __init__(self, dim=768, num_heads=8, dropout=0.0, gamma=0.6, bias=0.2) -> None:
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.temporal = Simple_SelfAttn(dim, num_heads, dropout)
        self.mlp = nn.Sequential(nn.Linear(dim, dim), nn.ReLU(), nn.Linear(dim, dim))
        self._flat: Optional[torch.Tensor] = None
        self._flat_grad: Optional[torch.Tensor] = None
        # inference-time GEMM precision (not a constructor argument: the reference's signature stays): "fp32" (default - exact fp32
        # MFMA on the parameter VALUES, the mode of every greedy-index parity claim) or "bf16" (the reference's own inference
        # precision, gen_id_tspo.py:55: bf16 GEMM operands, fp32 accumulation - ops.selector_forward(precision="bf16")).
        # Training (grad mode) always runs exact fp32 unless PolicyTrainer(gemm_precision="bf16x3") is used.
        self.inference_gemm_precision = "fp32"
        self._cache_key = None
        self._cache_flat = None

    # ---- reference utilities (host side; not used by forward) ----------------
    def create_causal_mask(self, seq_len):
        return torch.tril(torch.ones(seq_len, seq_len))

    def create_window_mask(self, seq_len, window_size=8):
        """[T,T] 0/1 mask of the window *set* (temporal_agent.py:97-104), vectorised."""
        j = torch.arange(seq_len).unsqueeze(1)
        c = torch.arange(seq_len).unsqueeze(0)
        lo = (j - window_size // 2).clamp(min=0)
        hi = (j - window_size // 2 + window_size - 1).clamp(max=seq_len - 1)
        return ((c >= lo) & (c <= hi)).float()

    def pair_cosine(self, a, b):
        cos_sim = torch.einsum('bnc,bmc->bnm', a, b)
        a_norm = torch.sqrt((a ** 2).sum(dim=-1)).unsqueeze(-1)
        b_norm = torch.sqrt((b ** 2).sum(dim=-1)).unsqueeze(1)
        return cos_sim / (a_norm * b_norm + 1e-6)

    # ---- parameter bucket -----------------------------------------------------
    def _named(self) -> Dict[str, nn.Parameter]:
        return dict(self.named_parameters())

    def _trainable_named(self):
        named = self._named()
        return [(n, named[n]) for n, _ in ops.FLAT_LAYOUT if "ffn_o" not in n]

    def flatten_parameters(self, device=None) -> torch.Tensor:
        """Move all 12 tensors into one contiguous fp32 bucket (views) + a matching grad bucket."""
        named = self._named()
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        offs = ops.flat_offsets(self.dim)
        flat = torch.empty(offs["__total__"][0], dtype=torch.float32, device=dev)
        grad = torch.zeros_like(flat)
        for name, _ in ops.FLAT_LAYOUT:
            off, shape = offs[name]
            p = named[name]
            n = p.numel()
            flat[off:off + n].copy_(p.detach().to(device=dev, dtype=torch.float32).flatten())
            p.data = flat[off:off + n].view(shape)
            p.grad = grad[off:off + n].view(shape)
        self._flat, self._flat_grad = flat, grad
        return flat

    def _is_flat(self) -> bool:
        if self._flat is None:
            return False
        offs = ops.flat_offsets(self.dim)
        base = self._flat.data_ptr()
        named = self._named()
        return all(named[n].data_ptr() == base + 4 * offs[n][0] and named[n].dtype == torch.float32
                   for n, _ in ops.FLAT_LAYOUT)

    def _flat_params(self, training: bool = False) -> torch.Tensor:
        if self._is_flat():
            return self._flat
        # parameters were replaced (from_pretrained / .to(dtype) / load_state_dict): a packed fp32 copy is built.
        # While training (grad mode on and a trainable parameter) it is rebuilt on EVERY call: optimizers that write
        # through `p.data` (DeepSpeed / bf16 master-weight copies, EMA swaps) change neither data_ptr nor _version, so no
        # cache key can see them; the copy is 14 MB.  For inference the copy is cached, keyed on (data_ptr, _version,
        # dtype) of every tensor; `invalidate_packed()` drops it after an out-of-band in-place update.
        named = self._named()
        training = training or (torch.is_grad_enabled() and any(p.requires_grad for p in named.values()))
        key = tuple((named[n].data_ptr(), named[n]._version, named[n].dtype) for n, _ in ops.FLAT_LAYOUT)
        if training or key != self._cache_key or self._cache_flat is None:
            dev = next(self.parameters()).device
            offs = ops.flat_offsets(self.dim)
            flat = self._cache_flat
            if flat is None or flat.device != dev:
                flat = torch.empty(offs["__total__"][0], dtype=torch.float32, device=dev)
            for n, _ in ops.FLAT_LAYOUT:
                off, _shape = offs[n]
                flat[off:off + named[n].numel()].copy_(named[n].detach().flatten())
            self._cache_key, self._cache_flat = key, flat
        return self._cache_flat

    def invalidate_packed(self) -> None:
        """Forget the cached packed copy of non-flat parameters (call after writing weights through `p.data`)."""
        self._cache_key, self._cache_flat = None, None

    def _realias_grads(self) -> None:
        """Flat mode: every p.grad must be a view of the gradient bucket (`optimizer.zero_grad(set_to_none=True)` or
        `module.zero_grad()` drop the views; autograd would then allocate fresh grads outside the bucket and the bucket the
        data-parallel trainer all-reduces would silently go stale)."""
        if self._flat is None or not self._is_flat():
            return
        offs = ops.flat_offsets(self.dim)
        base = self._flat_grad.data_ptr()
        for n, p in self._named().items():
            off, shape = offs[n]
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                view = self._flat_grad[off:off + p.numel()].view(shape)
                if p.grad is not None:
                    view.copy_(p.grad)
                else:
                    view.zero_()
                p.grad = view

    def zero_grad(self, set_to_none: bool = False):
        """Flat mode keeps the gradient views alive: the bucket is zeroed in place whatever `set_to_none` says."""
        if self._flat is not None and self._is_flat():
            self._flat_grad.zero_()
            self._realias_grads()
            return
        return super().zero_grad(set_to_none=set_to_none)

    # ---- forward --------------------------------------------------------------
    def forward(self, input_emb, text_emb, clip_scores=None, window_size=None, score_tau=0.025):
        """input_emb [T,d], text_emb [1,d] | [M,d] | [1,M,d], clip_scores [T]
        -> (sim_total [T], temporal_attn [1,T,d]) in input_emb's dtype."""
        T, D = input_emb.shape
        if window_size is None:   # reference: `None // 2` inside create_window_mask
            raise TypeError("unsupported operand type(s) for //: 'NoneType' and 'int'")
        if clip_scores is None:   # reference: `sim_total + None`
            raise TypeError("unsupported operand type(s) for +: 'Tensor' and 'NoneType'")
        if D != self.dim:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({T}x{D} and {self.dim}x{self.dim})")
        txt = text_emb if text_emb.ndim == 3 else text_emb.unsqueeze(0)
        scores, attn = self.forward_batched(input_emb.unsqueeze(0), txt, clip_scores.reshape(1, T), window_size,
                                            score_tau)
        return scores[0].to(input_emb.dtype), attn.to(input_emb.dtype)

    def forward_batched(self, img, txt, clip, window_size, score_tau=0.025):
        """img [B,T,D], txt [B,M,D], clip [B,T] -> (scores f32 [B,T], temporal_attn f32 [B,T,D])."""
        params = [p for _, p in self._trainable_named()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            self._realias_grads()
            return _SelectorFn.apply(self, img, txt, clip, int(window_size), float(score_tau), *params)
        scores, attn, _ = ops.selector_forward(self._flat_params(), img, txt, clip, self.num_heads, int(window_size),
                                               float(score_tau), precision=self.inference_gemm_precision)
        return scores, attn


def _image_features(clip_model, pixel_values):
    """HIP CLIP-L encode with weights packed from the HF module's state.  The packed bf16 copy is cached on the module,
    keyed on (data_ptr, _version) of EVERY vision-tower tensor and the device: load_state_dict / an in-place update /
    .to(device) rebuild it.  (Writes through `p.data` are invisible to any key: call `invalidate_packed_clip(model)`.)"""
    packed = getattr(clip_model, "_tspo_packed", None)
    vm = clip_model.vision_model
    tensors = list(vm.parameters()) + [clip_model.visual_projection.weight]
    key = (tensors[0].device, tuple((t.data_ptr(), t._version) for t in tensors))
    if packed is None or packed[0] != key:
        cfg = clip_model.config.vision_config
        c = dict(hidden=cfg.hidden_size, layers=cfg.num_hidden_layers, heads=cfg.num_attention_heads,
                 mlp=cfg.intermediate_size, patch=cfg.patch_size, image=cfg.image_size,
                 proj=clip_model.config.projection_dim, ln_eps=cfg.layer_norm_eps)
        state = {"vision_model." + k: v for k, v in vm.state_dict().items()}
        state["visual_projection.weight"] = clip_model.visual_projection.weight
        packed = (key, ops.ClipVitWeights(state, c, key[0]))
        clip_model._tspo_packed = packed
    return ops.clip_vit_forward(packed[1], pixel_values)


def invalidate_packed_clip(clip_model) -> None:
    clip_model._tspo_packed = None


def _flash_attn_available() -> bool:
    import importlib.util
    return importlib.util.find_spec("flash_attn") is not None


def _pooled(out):
    """transformers >= 5 returns BaseModelOutputWithPooling from get_*_features; 4.49 (the
    reference's pin) returns the projected tensor."""
    return out if isinstance(out, torch.Tensor) else out.pooler_output


def _frames_to_pil(candidates, processor_type):
    import PIL.Image as Image
    image_list = []
    for j in range(len(candidates)):
        if processor_type == 'llava':
            raw_image = np.array(candidates[j])
        else:
            raw_image = candidates[j].permute(1, 2, 0).cpu().numpy().astype(np.uint8)
        image_list.append(Image.fromarray(raw_image))
    return image_list


def _frames_to_device_u8(candidates, processor_type, dev, clip_processor):
    """uint8 frames -> GPU -> Pillow-exact resize/crop in HIP (tspo_preprocess_frames) -> uint8 [T,3,224,224]; the
    normalisation is fused into the encoder's patch gather.  Returns None when the processor is not configured like
    CLIP's default one or the frames are not a uniform uint8 batch (then the PIL path of the reference is used)."""
    from .preprocess import preprocess_frames, processor_is_default_clip
    ip = getattr(clip_processor, "image_processor", None)
    if ip is None or not processor_is_default_clip(ip):
        return None
    try:
        if processor_type == 'llava':
            arr = candidates if isinstance(candidates, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(candidates)))
        else:
            arr = torch.stack([c for c in candidates]) if not isinstance(candidates, torch.Tensor) else candidates
            arr = arr.to(torch.uint8) if arr.dtype != torch.uint8 else arr
        if arr.dtype != torch.uint8 or arr.ndim != 4:
            return None
        return preprocess_frames(arr.to(dev))
    except (ValueError, TypeError, RuntimeError):
        return None


_PIL_FALLBACK_WARNED = False


def extract_clip_features_impl(clip_model, clip_processor, candidates, problem, processor_type='llava'):
    """temporal_agent.py:151-169 / utils.py:18-35 / tspo_trainer.py:387-404: text tower on stock
    PyTorch-ROCm, all T frames through the HIP CLIP encoder, cosine clip score in HIP."""
    dev = next(clip_model.parameters()).device
    inputs_text = clip_processor(text=problem, return_tensors="pt", padding=True, truncation=True).to(dev)
    with torch.no_grad():
        text_features = _pooled(clip_model.get_text_features(**inputs_text))
    pixels = _frames_to_device_u8(candidates, processor_type, dev, clip_processor)
    if pixels is None:   # non-default processor config / ragged frames: the reference's own CPU PIL path
        global _PIL_FALLBACK_WARNED
        if not _PIL_FALLBACK_WARNED:      # said once per process: the caller has lost the on-device front end
            _PIL_FALLBACK_WARNED = True
            import warnings
            warnings.warn("tspo_amd: frames are preprocessed on the CPU by the stock CLIP processor (PIL, one frame at a time, as "
                          "the reference does) - the on-device resize / crop (tspo_preprocess_frames) applies only to uniform uint8 "
                          "frame batches and a processor configured like CLIP's default one; results are unchanged, the front end is "
                          "~100x slower", RuntimeWarning, stacklevel=2)
        image_list = _frames_to_pil(candidates, processor_type)
        pixels = clip_processor(images=image_list, return_tensors="pt", padding=True).to(dev)["pixel_values"]
    with torch.no_grad():
        image_features = _image_features(clip_model, pixels)
        clip_scores = ops.clip_scores(text_features[:1].float().unsqueeze(0), image_features.unsqueeze(0))[0]
    dt = clip_model.dtype
    return image_features.to(dt), text_features, clip_scores.to(dt)


try:  # transformers is present in this image; keep the import local so the ops layer does not depend on it
    from transformers import CLIPModel
except Exception:  # pragma: no cover
    CLIPModel = nn.Module  # type: ignore


class TSPOModel(CLIPModel):
    """CLIP-L + selector ("TSPO-0.4B"), reference API (temporal_agent.py:146-231)."""

    def __init__(self, clip_config):
        super().__init__(clip_config)
        self.selector = MultiModal_Align()

    def extract_feature(self, clip_processor, candidates, problem, processor_type='llava'):
        return extract_clip_features_impl(self, clip_processor, candidates, problem, processor_type)

    def extract_feature_from_pixels(self, pixel_values, text_features):
        """Fast path for already-preprocessed frames [T,3,224,224] (any of f32/bf16/f16/uint8) on the GPU."""
        with torch.no_grad():
            image_features = _image_features(self, pixel_values)
            clip_scores = ops.clip_scores(text_features[:1].float().unsqueeze(0), image_features.unsqueeze(0))[0]
        return image_features, text_features, clip_scores

    def temporal_sampling(self, image_features, text_features, clip_scores, method, window_size, sample_num):
        pred_score, _ = self.selector(image_features, text_features, clip_scores, window_size=window_size)
        ts_ids, _ = self.inference_ts(pred_score, method=method, sample_len=sample_num)
        return ts_ids, pred_score

    def forward(self, clip_processor, candidates, problem, sample_num, window_size=12, method='topk',
                processor_type='llava'):
        image_features, text_features, clip_scores = self.extract_feature(clip_processor, candidates, problem,
                                                                          processor_type)
        ts_ids, pred_score = self.temporal_sampling(image_features, text_features, clip_scores, method, window_size,
                                                    sample_num)
        return ts_ids, pred_score

    def inference_ts(self, confidence, method, sample_len):
        return inference_ts(confidence, method, sample_len)

    @classmethod
    def from_merged_components(cls, clip_model, selector_state_dict: dict, **kwargs):
        config = clip_model.config
        model = cls(config).to(clip_model.device).to(clip_model.dtype)
        model.load_state_dict(clip_model.state_dict(), strict=False)
        model.selector.load_state_dict(selector_state_dict)
        return model

    def save_pretrained(self, save_directory: str, **kwargs):
        super().save_pretrained(save_directory, **kwargs)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        """HF entry point exactly as the reference's callers use it (mp_tools/vlmeval/vlm/gen_id_tspo.py:55:
        `from_pretrained(path, attn_implementation="flash_attention_2", torch_dtype=torch.bfloat16, device_map="auto")`).
        `flash_attention_2` only ever applied to the two stock CLIP towers; here the vision tower's attention is the HIP
        kernel whatever this says, and the <= 77-token text tower runs on stock PyTorch-ROCm: when the flash_attn
        package is not installed (it is not part of the ROCm image) the request maps to PyTorch SDPA instead of raising."""
        if kwargs.get("attn_implementation") == "flash_attention_2" and not _flash_attn_available():
            kwargs["attn_implementation"] = "sdpa"
        return super().from_pretrained(pretrained_model_name_or_path, *model_args, **kwargs)


def inference_ts(confidence, method, sample_len):
    """temporal_agent.py:187-214 / llava_qwen.py:146-176.  'topk' and 'bin-max' run on the GPU
    (tspo_topk_sorted / tspo_binmax); 'aks' is host numpy exactly as in the reference."""
    print(f"sample_method: {method}")
    if method == "topk":
        sel_idx = ops.topk_sorted(confidence, sample_len)
    elif method == "bin-max":
        sel_idx = ops.binmax(confidence, sample_len)
    elif method == "aks":
        sel_idx = AKS_sampling(confidence.float().cpu().numpy(), sample_len)
        sel_idx = torch.tensor(sel_idx).cuda()
    else:
        raise UnboundLocalError("local variable 'sel_idx' referenced before assignment")  # reference behaviour
    return sel_idx, confidence
