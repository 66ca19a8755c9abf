"""GPU tests of the flows either side of the hot path: HF entry points on the HIP-backed TSPOModel, the
evaluation harness's generate_inner flow (cache miss / hit, bf16, T <= 64 skip, VideoMME -> bin-max), weight caches
under in-place updates, the needle-in-haystack builder feeding the temporal reward in a policy step, and the training
driver (gradient accumulation, LR decay, tau annealing, JSONL metrics, checkpoint + resume)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import tspo_oracle as O
from tspo_amd import haystack, io as tio, ops, rewards, synth
from tspo_amd.pipeline import PolicyTrainer
from tspo_amd.temporal_agent import MultiModal_Align, TSPOModel, invalidate_packed_clip
from test_gpu_api import _StubProcessor

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T_(x):
    return torch.from_numpy(np.asarray(x))


def _tiny_cfg():
    from transformers import CLIPConfig
    return CLIPConfig(text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                                       vocab_size=49408, max_position_embeddings=77, projection_dim=768),
                      vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                         image_size=224, patch_size=14, projection_dim=768), projection_dim=768)


def _tiny_model(seed=0):
    torch.manual_seed(seed)
    model = TSPOModel(_tiny_cfg()).float().eval()
    for p in model.vision_model.parameters():
        if p.ndim >= 2:
            torch.nn.init.normal_(p, std=0.05)
    model.selector.load_state_dict({k: T_(v) for k, v in synth.selector_state(768, seed=2, std=0.02).items()})
    return model


def test_from_pretrained_roundtrip_forward_on_gpu(tmp_path):
    """save_pretrained -> from_pretrained(attn_implementation="flash_attention_2", torch_dtype=bf16, device_map="auto")
    (gen_id_tspo.py:55): the reloaded model sits on the GPU in bf16 and `forward` returns exactly what the model that was
    saved returns."""
    model = _tiny_model().to(torch.bfloat16).to(DEV)
    frames = synth.uniform_u8((72, 240, 320, 3), 91)
    proc = _StubProcessor()
    ids0, pred0 = model(proc, frames, "what is shown?", sample_num=8, window_size=12, method="topk")
    model.save_pretrained(str(tmp_path / "m"))
    back = TSPOModel.from_pretrained(str(tmp_path / "m"), attn_implementation="flash_attention_2", torch_dtype=torch.bfloat16,
                                     device_map="auto")
    assert back.device.type == "cuda" and back.dtype == torch.bfloat16
    ids1, pred1 = back(proc, frames, "what is shown?", sample_num=8, window_size=12, method="topk")
    assert torch.equal(ids0, ids1) and torch.equal(pred0, pred1)
    assert ids1.dtype == torch.int64 and pred1.dtype == torch.bfloat16 and pred1.shape == (72,)


class _FakeVideo:
    """decord.VideoReader stand-in: n frames of 120x160 RGB at 30 fps, frame i a pure function of (seed, i)."""

    def __init__(self, n, seed):
        self.n, self.seed = n, seed

    def __len__(self):
        return self.n

    def get_avg_fps(self):
        return 30.0

    def get_batch(self, idx):
        arr = np.stack([synth.uniform_u8((120, 160, 3), self.seed * 100003 + int(i)) for i in idx])

        class _B:
            def asnumpy(self_inner):
                return arr
        return _B()


@pytest.mark.parametrize("dataset", ["MLVU", "VideoMME"])
def test_reproduce_published_idx_flow_round_trip(tmp_path, dataset):
    """tools/reproduce_published_idx.py `run()` - the one command from local weights + videos to the comparison with a published
    frame list - end to end on the GPU with what this image CAN provide: a small random TSPOModel checkpoint (save_pretrained),
    the stub processor, an in-memory video reader, and a fake reference checkout holding the annotation docs.  Pass 1 produces the
    frame-index JSON (feature-cache misses); it is then installed as the 'published' file and pass 2 reproduces it from the caches:
    exact match on every doc, the >64-candidate docs selected (top-k / bin-max), the short one listing all its candidates."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, os.path.join(root, "tools"))
    spec = importlib.util.spec_from_file_location("reproduce_published_idx", os.path.join(root, "tools", "reproduce_published_idx.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    fname, key, video_of = tool.ANNO[dataset]
    ref = tmp_path / "ref"
    (ref / "evaluation" / "jsons").mkdir(parents=True)
    (ref / "evaluation" / "jsons_idx").mkdir(parents=True)
    mk = (lambda i, q: {"question_id": f"Q{i}", "video_name": f"v{i}.mp4", "question": q + "\n(A) yes\n(B) no", "answer": "A"}) if dataset == "MLVU" \
        else (lambda i, q: {"question_id": f"00{i}-1", "videoID": f"v{i}", "question": q, "options": ["A. yes", "B. no"], "answer": "A"})
    docs = [mk(0, "what happens first?"), mk(1, "who enters the room?"), mk(2, "is it short?")]
    json.dump(docs, open(ref / "evaluation" / "jsons" / fname, "w"))
    json.dump([], open(ref / "evaluation" / "jsons_idx" / f"TSPO_{dataset}_frameIdx.json", "w"))
    lengths = {"v0.mp4": 30 * 90, "v1.mp4": 30 * 70, "v2.mp4": 30 * 40}           # 90 / 70 / 40 one-fps candidates
    opened = []

    def open_video(path):
        opened.append(os.path.basename(path))
        return _FakeVideo(lengths[os.path.basename(path)], len(os.path.basename(path)) + int(os.path.basename(path)[1]))
    _tiny_model().to(torch.bfloat16).save_pretrained(str(tmp_path / "w"))
    kw = dict(dataset=dataset, weights=str(tmp_path / "w"), videos=str(tmp_path / "videos"), reference=str(ref),
              save_root=str(tmp_path / "feats"), processor=_StubProcessor(), open_video=open_video)
    s1 = tool.run(out=str(tmp_path / "pass1.json"), **kw)
    assert s1["cache_misses"] == 3 and s1["cache_hits"] == 0 and s1["docs_compared"] == 0 and len(opened) == 3
    produced = json.load(open(tmp_path / "pass1.json"))
    by = {d[key]: d["frame_idx"] for d in produced}
    k0, k1, k2 = (d[key] for d in docs)
    assert len(by[k0]) == 64 and len(by[k1]) == 64 and by[k2] == [float(30 * i) for i in range(40)]      # T <= 64: every candidate
    assert all(b > a for a, b in zip(by[k0], by[k0][1:])) and all(v % 30 == 0 for v in by[k0])
    os.replace(tmp_path / "pass1.json", ref / "evaluation" / "jsons_idx" / f"TSPO_{dataset}_frameIdx.json")
    s2 = tool.run(out=str(tmp_path / "pass2.json"), **kw)
    assert s2["cache_hits"] == 3 and len(opened) == 3, "the second pass must come from the feature caches"
    assert s2["docs_compared"] == 3 and s2["exact_match_rate"] == 1.0 and s2["missing_in_produced"] == 0 and s2["mean_jaccard"] == 1.0


@pytest.mark.parametrize("dataset", ["VideoMME", "MLVU"])
def test_generate_inner_flow_cache_miss_and_hit(tmp_path, dataset):
    """mp_tools/vlmeval/vlm/gen_id_tspo.py:59-92 with the real HIP TSPOModel (bf16, as the harness loads it): miss ->
    extract_feature -> .pth cache; hit -> same ids; the float frame numbers == the oracle's selection (bin-max for VideoMME,
    top-k otherwise) on the cached features; videos of <= 64 frames are returned whole."""
    model = _tiny_model(1).to(torch.bfloat16).to(DEV)
    T = 150
    frames = synth.uniform_u8((T, 224, 224, 3), 17)
    sampled_idx = torch.arange(0, 30 * T, 30)                       # absolute frame numbers at 1 fps of a 30 fps video
    calls = []

    def load_video(path, max_frames_num, fps, force_sample):        # decord stand-in
        calls.append(path)
        n = 40 if "short" in path else T
        return frames[:n], None, None, sampled_idx[:n]

    gen = tio.FrameIdGenerator(model, _StubProcessor(), str(tmp_path), sample_num=64, load_video=load_video)
    msg = [{"value": "long.mp4"}, {"value": "<image>\nQuestion: what is shown?\nOptions:\nA. a\nB. b"}]
    ids_miss = gen.generate_inner(msg, index=7, dataset=dataset)
    assert gen.cache_misses == 1 and calls == ["long.mp4"]
    stat = torch.load(tio.feature_cache_path(str(tmp_path), dataset, 7))
    assert sorted(stat) == ["clip_scores", "image", "sampled_idx", "text"] and stat["image"].dtype == torch.bfloat16
    assert stat["image"].shape == (T, 768) and stat["image"].device.type == "cpu"
    ids_hit = gen.generate_inner(msg, index=7, dataset=dataset)
    assert gen.cache_hits == 1 and calls == ["long.mp4"] and ids_hit == ids_miss
    assert len(ids_hit) == 64 and all(isinstance(x, float) for x in ids_hit) and ids_hit == sorted(ids_hit)
    # oracle selection on the cached (bf16) features
    sel = {k: v.detach().float().cpu() for k, v in model.selector.state_dict().items()}
    s_ref, _ = O.selector_forward(sel, stat["image"].float(), stat["text"].float(), stat["clip_scores"].float(), 12, 0.025)
    s_ref = s_ref.to(torch.bfloat16).float()                        # the model returns scores in its dtype (bf16)
    pick = O.binmax(s_ref, 64) if dataset == "VideoMME" else O.topk_sorted(s_ref, 64)
    want = sampled_idx[pick].float().tolist()
    got_scores = model.temporal_sampling(stat["image"].to(DEV), stat["text"].to(DEV), stat["clip_scores"].to(DEV),
                                         "topk", 12, 64)[1].float().cpu()
    if torch.equal(got_scores, s_ref):                               # identical bf16 scores -> identical selection
        assert ids_hit == want
    else:                                                            # (bf16 rounding of near-equal scores may differ by 1 ulp)
        assert len(set(ids_hit) & set(want)) >= 60
    # T <= sample_num: no selection, the sampled frame numbers themselves
    short = gen.generate_inner([{"value": "short.mp4"}, msg[1]], index=8, dataset=dataset)
    assert short == sampled_idx[:40].float().tolist()


def test_weight_caches_see_in_place_updates():
    m = MultiModal_Align(dim=64, num_heads=8).to(DEV)
    m.load_state_dict({k: T_(v) for k, v in synth.selector_state(64, seed=4, std=0.1, bias_std=0.05).items()})
    img, txt, clip = (T_(synth.normal(s, i)).to(DEV) for i, s in enumerate([(40, 64), (1, 64), (40,)], 70))
    with torch.no_grad():
        s0, _ = m(img, txt, clip, window_size=12)
    # training mode (grad enabled): a write through p.data (what DeepSpeed / EMA do) is seen by the very next forward
    m.mlp[2].weight.data.mul_(0.0)
    s1, _ = m(img, txt, clip, window_size=12)
    assert not torch.allclose(s1.detach(), s0)
    # inference mode caches the packed copy: out-of-band writes need invalidate_packed(), versioned writes do not
    with torch.no_grad():
        s2, _ = m(img, txt, clip, window_size=12)
        assert torch.equal(s2, s1.detach())
        m.mlp[2].bias.data.add_(1.0)
        m.invalidate_packed()
        s3, _ = m(img, txt, clip, window_size=12)
        assert not torch.equal(s3, s2)
        m.mlp[2].bias.add_(1.0)                                    # in-place op on the parameter: bumps _version
        s4, _ = m(img, txt, clip, window_size=12)
        assert not torch.equal(s4, s3)
    # two graphs alive across a parameter update: the earlier graph's backward must use the weights ITS forward used
    m3 = MultiModal_Align(dim=64, num_heads=8).to(DEV)
    m3.load_state_dict({k: T_(v) for k, v in synth.selector_state(64, seed=4, std=0.1, bias_std=0.05).items()})
    ref = MultiModal_Align(dim=64, num_heads=8).to(DEV)
    ref.load_state_dict(m3.state_dict())
    sa, _ = m3(img, txt, clip, window_size=12)                      # graph A (non-flat parameters: packed copy)
    m3.mlp[0].weight.data.mul_(2.0)                                 # "optimizer step" through p.data
    sb, _ = m3(img, txt, clip, window_size=12)                      # graph B repacks the shared buffer in place
    sa.sum().backward()
    sr, _ = ref(img, txt, clip, window_size=12)
    sr.sum().backward()
    assert torch.equal(m3.mlp[2].weight.grad, ref.mlp[2].weight.grad)    # A's gradients: the pre-update weights
    assert not torch.equal(sb.detach(), sa.detach())
    mf = MultiModal_Align(dim=64, num_heads=8).to(DEV)
    mf.load_state_dict(ref.state_dict())
    mf.flatten_parameters()
    sf, _ = mf(img, txt, clip, window_size=12)
    mf._flat.mul_(1.5)                                              # in-place update of the live bucket before the backward
    with pytest.raises(RuntimeError, match="modified in place"):
        sf.sum().backward()
    # flat mode: gradient views survive zero_grad(set_to_none=True) and an optimizer's zero_grad()
    m2 = MultiModal_Align(dim=64, num_heads=8).to(DEV)
    m2.load_state_dict({k: T_(v) for k, v in synth.selector_state(64, seed=4, std=0.1, bias_std=0.05).items()})
    m2.flatten_parameters()
    opt = torch.optim.SGD(m2.parameters(), lr=0.1)
    for zero in (lambda: m2.zero_grad(set_to_none=True), lambda: opt.zero_grad(set_to_none=True)):
        zero()
        s, _ = m2(img, txt, clip, window_size=12)
        s.sum().backward()
        base, n = m2._flat_grad.data_ptr(), m2._flat_grad.numel()
        for name, p in m2.named_parameters():
            assert p.grad is not None and base <= p.grad.data_ptr() < base + 4 * n, name
        assert float(m2._flat_grad.abs().sum()) > 0
    # packed CLIP weights follow load_state_dict on the vision tower
    model = _tiny_model(3).to(DEV)
    px = T_(synth.uniform_u8((4, 3, 224, 224), 5)).to(DEV)
    f0 = model.extract_feature_from_pixels(px, torch.zeros(1, 768, device=DEV))[0].clone()
    sd = {k: v * 0.5 for k, v in model.vision_model.state_dict().items()}
    model.vision_model.load_state_dict(sd)
    f1 = model.extract_feature_from_pixels(px, torch.zeros(1, 768, device=DEV))[0]
    assert not torch.allclose(f0, f1)
    invalidate_packed_clip(model)
    assert torch.equal(model.extract_feature_from_pixels(px, torch.zeros(1, 768, device=DEV))[0], f1)


def test_policy_step_on_needle_in_haystack_sample():
    """tspo_trainer.py:462-482 + :554-573 end to end on device: true clips of the question's video mixed with distractor
    clips (seeded like the reference), frames -> HIP CLIP features, G rollouts, temporal reward from `shuffle_mask`
    (batched on the GPU == the reference's per-rollout host loop), specific-type reward combination, one policy step."""
    model = _tiny_model(5).to(DEV)
    L, G, k = 64, 4, 8
    video = T_(synth.uniform_u8((L, 224, 224, 3), 31)).to(DEV)
    wrong = haystack.synthetic_distractors(3, 20, video, seed=9)
    mixed, mask = haystack.build_specific_sample(video, wrong, repeat_times=2, sample_len=20, rng=np.random.RandomState(3))
    assert mixed.is_cuda and mixed.shape == (100, 224, 224, 3) and int(mask.sum()) == 40
    feats, text, clip = model.extract_feature(_StubProcessor(), mixed, "where is the cat?")
    flat = model.selector.flatten_parameters()
    tr = PolicyTrainer(flat, dim=768, heads=8, window_size=12)
    f, t, c = feats.float()[None], text.float()[:1][None], clip.float()[None]
    scores, idx, logp, ctx = tr.rollout(f, t, c, G, k, 0.025)
    temporal = rewards.selection_mask_reward_gpu(idx, mask.to(DEV)[None])
    host = rewards.temporal_localization_reward(None, None, [(i, i) for i in idx[0]], mask)
    np.testing.assert_allclose(temporal[0].cpu().numpy(), np.array(host), atol=1e-7)
    acc = (temporal > 0.4).float()
    rew = rewards.combine_rewards(torch.stack([acc, temporal], -1).reshape(G, 2), "specific").reshape(1, G)
    before = flat.clone()
    st = tr.update(ctx, f, t, logp, idx, rew)
    assert torch.isfinite(tr.flat).all() and "grad_norm_scale" in st
    if float(rew.std()) > 0:
        assert not torch.equal(tr.flat, before)
    adv_ref = O.grpo_advantage(rew.cpu().flatten(), G)
    np.testing.assert_allclose(st["advantages"].cpu().flatten().numpy(), adv_ref.numpy(), rtol=1e-5, atol=1e-6)


def test_training_driver_accumulation_logging_checkpoint_resume(tmp_path):
    from tspo_amd import train as tt
    kw = dict(max_steps=6, num_generations=4, training_sample_len=8, gradient_accumulation_steps=2, save_steps=2,
              save_total_limit=2, dim=64, heads=8, seed=11)
    data = lambda: tt.SyntheticFeatures(T=96, D=64, seed=11, device=DEV, signal=0.1)   # weak signal: rewards differ, gradients flow
    cfg_a = tt.TrainConfig(output_dir=str(tmp_path / "a"), **kw)
    ma = tt.train(cfg_a, data(), resume=False)
    lines = [json.loads(l) for l in open(os.path.join(cfg_a.output_dir, "metrics.jsonl"))]
    assert [l["step"] for l in lines] == [1, 2, 3, 4, 5, 6]
    assert max(l["grad_norm"] for l in lines) > 0 and max(l["reward_std"] for l in lines) > 0   # the run actually learns something
    assert abs(lines[0]["learning_rate"] - 5e-4) < 1e-12 and abs(lines[3]["learning_rate"] - 5e-4 * 3 / 6) < 1e-12   # linear decay
    assert abs(lines[2]["score_tau"] - (0.025 - (0.025 - 0.01) / 6 * 2)) < 1e-12                                    # tau annealing
    for key in ("reward", "reward_std", "advantages", "ts_length", "rewards/accuracy_reward",
                "rewards/temporal_localization_reward", "loss", "grad_norm"):
        assert key in lines[-1], key
    assert sorted(os.listdir(cfg_a.output_dir)) == ["checkpoint-4", "checkpoint-6", "metrics.jsonl"]                # save_total_limit 2
    ck = tio.load_selector_safetensors(os.path.join(cfg_a.output_dir, "checkpoint-6", "model.safetensors"))
    offs = ops.flat_offsets(64)
    for name, (o, shape) in offs.items():
        if not name.startswith("__"):
            assert torch.equal(ck[name].to(DEV), ma["flat"][o:o + int(np.prod(shape))].view(shape)), name
    # interrupted after step 4, resumed: identical parameters to the uninterrupted run
    cfg_b = tt.TrainConfig(output_dir=str(tmp_path / "b"), **kw)
    tt.train(cfg_b, data(), resume=False, stop_after=4)
    mb = tt.train(cfg_b, data(), resume=True)
    assert torch.equal(mb["flat"], ma["flat"])
