"""The oracle (oracle/tspo_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from inputs import (SELECTOR_CASES, GUMBEL_CASES, TRAIN_CASES, CLIP_CASES, selector_inputs, gumbel_logits,
                    train_inputs, clip_pixels)
from oracle import tspo_oracle as O
from tspo_amd import synth


def T_(x):
    return torch.from_numpy(np.asarray(x))


def state_t(state):
    return {k: T_(v) for k, v in state.items()}


def test_positional_encoding(golden):
    g = golden["misc"]
    np.testing.assert_array_equal(O.positional_encoding(4, 8).numpy(), g["pe_4_8"])
    np.testing.assert_array_equal(O.positional_encoding(5, 6).numpy(), g["pe_5_6"])


def test_generate_uniform_integers(golden):
    g = golden["misc"]
    for key in g.files:
        if key.startswith("uni_"):
            _, t, l = key.split("_")
            assert O.generate_uniform_integers(int(t), int(l)) == g[key].tolist()


@pytest.mark.parametrize("case", SELECTOR_CASES, ids=[c[0] for c in SELECTOR_CASES])
def test_selector_forward(golden, case):
    name, T, D, H, w, tau, M, ks = case
    g = golden["selector"]
    img, txt, clip, state = selector_inputs(name, T, D, M)
    s, h = O.selector_forward(state_t(state), T_(img), T_(txt), T_(clip), w, tau, num_heads=H)
    # same ops as the reference on the same machine -> tight
    np.testing.assert_allclose(s.numpy(), g[f"{name}.scores"], rtol=1e-5, atol=2e-4)
    if f"{name}.attn" in g.files:
        np.testing.assert_allclose(h.numpy(), g[f"{name}.attn"], rtol=1e-5, atol=1e-5)
    else:
        np.testing.assert_allclose(h[0, [0, 1, T // 2, T - 1]].numpy(), g[f"{name}.attn_rows"], rtol=1e-5, atol=1e-5)
    if f"{name}.mask" in g.files:
        np.testing.assert_array_equal(O.create_window_mask(T, w).numpy().astype(np.uint8), g[f"{name}.mask"])


@pytest.mark.parametrize("case", SELECTOR_CASES, ids=[c[0] for c in SELECTOR_CASES])
def test_inference_ts(golden, case):
    name, T, D, H, w, tau, M, ks = case
    g = golden["selector"]
    s = T_(g[f"{name}.scores"])          # reference's own scores -> index parity is exact
    for k in ks:
        np.testing.assert_array_equal(O.topk_sorted(s, k).numpy(), g[f"{name}.topk{k}"])
        np.testing.assert_array_equal(O.binmax(s, k).numpy(), g[f"{name}.binmax{k}"])
    for k in (8, 16):
        if f"{name}.aks{k}" in g.files:
            assert O.aks_sampling(s.numpy(), k) == g[f"{name}.aks{k}"].tolist()


def test_topk_tie_rule(golden):
    g = golden["misc"]
    s = T_(g["ties.scores"])
    idx = O.topk_sorted(s, 3)
    assert idx.tolist() == [1, 2, 4]                      # declared: lowest indices among equal values
    np.testing.assert_array_equal(np.sort(s[idx].numpy()), np.sort(g["ties.top3_values"]))


@pytest.mark.parametrize("case", GUMBEL_CASES, ids=[c[0] for c in GUMBEL_CASES])
def test_gumbel_topk(golden, case):
    name, T, k, G, scale = case
    g = golden["gumbel"]
    logits = T_(gumbel_logits(T, scale))
    for i in range(G):
        idx, probs, logp = O.gumbel_topk(logits, T_(g[f"{name}.noise"][i]), k)
        np.testing.assert_array_equal(idx.numpy(), g[f"{name}.idx"][i])
        np.testing.assert_allclose(probs.numpy(), g[f"{name}.probs"][i], rtol=0, atol=1e-6)
        np.testing.assert_allclose(logp.numpy(), g[f"{name}.logp"], rtol=1e-6, atol=1e-5)


def test_gumbel_k_gt_T_raises():
    with pytest.raises(RuntimeError):
        O.gumbel_topk(torch.zeros(4), torch.zeros(4), 5)


def test_philox_known_answer():
    # Random123 KAT for philox4x32-10: ctr=0,key=0 and ctr=ff..,key=ff..
    z = O.philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = O.philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(int(v)) for v in f] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    n = O.gumbel_noise_philox(2, 3, 4096, seed=2024, offset=5)
    assert np.isfinite(n).all() and abs(n.mean() - 0.5772) < 0.02 and abs(n.std() - 1.2825) < 0.03


def test_advantage(golden):
    g = golden["train"]
    for nm in ("eq", "gen", "two", "bg"):
        a = O.grpo_advantage(T_(g[f"adv.{nm}.r"]), int(g[f"adv.{nm}.G"]))
        np.testing.assert_allclose(a.numpy(), g[f"adv.{nm}.a"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", TRAIN_CASES, ids=[c[0] for c in TRAIN_CASES])
def test_train_step(golden, case):
    name, T, D, H, w, tau, k, G = case
    g = golden["train"]
    img, txt, clip, state, rewards = train_inputs(name, T, D, G)
    np.testing.assert_array_equal(rewards, g[f"{name}.rewards"])
    idx, loss, grads = O.tspo_step_autograd(state_t(state), T_(img), T_(txt), T_(clip), T_(g[f"{name}.noise"]),
                                            T_(rewards), k, w, tau)
    np.testing.assert_array_equal(idx.numpy(), g[f"{name}.idx"])
    adv = O.grpo_advantage(T_(rewards), G)
    np.testing.assert_allclose(adv.numpy(), g[f"{name}.adv"], rtol=1e-6, atol=1e-6)
    assert abs(loss.item() - float(g[f"{name}.loss"])) < 1e-6
    # closed-form dL/dscores == reference autograd
    l2, ds = O.pg_grad_logits(T_(g[f"{name}.scores"]), idx, adv)
    np.testing.assert_allclose(ds.numpy(), g[f"{name}.dscores"], rtol=1e-4, atol=1e-7)
    assert abs(l2.item() - float(g[f"{name}.loss"])) < 1e-6
    for pn, gr in grads.items():
        if f"{name}.grad.{pn}" in g.files:
            ref = g[f"{name}.grad.{pn}"]
            np.testing.assert_allclose(gr.numpy(), ref, rtol=1e-4, atol=1e-6 * max(1.0, np.abs(ref).max()))
        else:
            ref = g[f"{name}.gradsl.{pn}"]
            np.testing.assert_allclose(gr.flatten()[:256].numpy(), ref, rtol=1e-4,
                                       atol=1e-5 * max(1e-6, np.abs(ref).max()))
            sums = g[f"{name}.gradsum.{pn}"]
            assert abs(gr.double().abs().sum().item() - sums[1]) <= 1e-4 * max(sums[1], 1e-12)
    # ffn_o never receives a gradient
    assert grads["temporal.ffn_o.weight"].abs().sum() == 0
    # AdamW step incl. clip-norm
    flat = torch.cat([grads[n].flatten() for n in O.SELECTOR_KEYS if "ffn_o" not in n])
    tn = flat.norm().item()
    assert abs(tn - float(g[f"{name}.gradnorm"])) < 1e-4 * max(1.0, tn)
    scale = O.clip_grad_scale(tn, 1.0)
    for pn in O.SELECTOR_KEYS:
        if "ffn_o" in pn:
            continue
        p0 = T_(state[pn])
        p1, _, _ = O.adamw_step(p0, grads[pn], torch.zeros_like(p0), torch.zeros_like(p0), 1, 5e-4,
                                grad_scale=scale)
        np.testing.assert_allclose(p1.flatten()[:256].numpy(), g[f"{name}.after.{pn}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["c2", "c4"])
def test_train_step_at_the_timed_sizes(golden, name):
    """The policy step at the sizes bench.py times (configs[2]: B=4, T=512, G=8, k=16; policy side of configs[4]: B=1, T=4096,
    G=16) - oracle vs the fixture `make_golden.py train_full` wrote from the reference's own modules: indices, advantages, losses,
    scores per prompt, the bucket (mean of the prompts' gradients), its norm and the parameters after clip + AdamW."""
    import _policy_full as P
    _, B, T, D, H, w, tau, k, G = P.CASES[name]
    g = golden["train_full"]
    img, txt, clip, state, rew = P.train_full_inputs(name, B, T, D, G)
    mean, per = P.oracle_mean_grads(name, g, range(B))
    for b, (idx, loss, _, adv, scores) in enumerate(per):
        np.testing.assert_array_equal(idx.numpy(), g[f"{name}.idx"][b])
        np.testing.assert_allclose(adv.numpy(), g[f"{name}.adv"][b], rtol=1e-6, atol=1e-6)
        assert abs(loss.item() - float(g[f"{name}.loss"][b])) < 2e-6
        np.testing.assert_allclose(scores.numpy(), g[f"{name}.scores"][b], rtol=1e-5, atol=2e-4)
    P.check_against_fixture(name, g, {n: v.numpy() for n, v in mean.items()}, f"oracle {name}", rtol=1e-4, atol_rel=1e-5)
    flat = torch.cat([mean[n].flatten() for n in P.TRAINED])
    tn = flat.norm().item()
    assert abs(tn - float(g[f"{name}.gradnorm"])) < 1e-4 * tn
    scale = O.clip_grad_scale(tn, 1.0)
    for pn in P.TRAINED:
        if pn == "temporal.Self_k.bias":
            continue      # zero gradient in exact arithmetic: Adam's first step there is lr x the sign of rounding noise
        p0 = T_(state[pn])
        p1, _, _ = O.adamw_step(p0, mean[pn], torch.zeros_like(p0), torch.zeros_like(p0), 1, 5e-4, grad_scale=scale)
        ok = np.abs(mean[pn].flatten()[:256].numpy()) * scale > 2e-6    # (the first Adam step is lr.g / (|g| + 1e-8): sign-like, ill-conditioned only where |g| ~ eps)
        np.testing.assert_allclose(p1.flatten()[:256].numpy()[ok], g[f"{name}.after.{pn}"][ok], rtol=1e-5, atol=1e-6)
    if name == "c2":
        # the one-forward form (used for T = 4096, where G dense graphs do not fit) == the reference-shaped 2.G-forward loop
        ps = state_t(state)
        i1, l1, g1, a1, s1 = O.tspo_step_autograd_one_forward(ps, T_(img[0]), T_(txt[0]), T_(clip[0]), T_(g["c2.noise"][0]),
                                                                T_(rew[0]), k, w, tau, H)
        i0, l0, g0, a0, s0 = per[0]
        assert torch.equal(i0, i1) and torch.equal(a0, a1) and abs(l0.item() - l1.item()) < 1e-6
        for pn in P.TRAINED:
            if pn == "temporal.Self_k.bias":
                continue
            assert (g0[pn] - g1[pn]).abs().max().item() <= 2e-5 * g0[pn].abs().max().item(), pn   # (fp32 order of the G-term sum)


@pytest.mark.parametrize("case", CLIP_CASES, ids=[c[0] for c in CLIP_CASES])
def test_clip_vit(golden, case):
    tag, cfg, n = case
    g = golden["clip"]
    w = {k: T_(v) for k, v in synth.clip_vision_state(**cfg).items()}
    _, px = clip_pixels(cfg, n)
    feats, hidden = O.clip_vit_forward(w, T_(px), num_heads=cfg["heads"], patch=cfg["patch"], return_hidden=True)
    ref = g[f"{tag}.feat"]
    np.testing.assert_allclose(feats.numpy(), ref, rtol=1e-4, atol=2e-5 * np.abs(ref).max())
    if tag == "clip_tiny":
        L = cfg["layers"]
        for i in range(1, L + 1):      # transformers' hidden_states[0] is pre-pre-LN; [i>=1] = after layer i
            np.testing.assert_allclose(hidden[i].numpy(), g[f"{tag}.hidden{i}"], rtol=1e-4, atol=1e-4)


def test_clip_normalize():
    u8, px = clip_pixels(synth.CLIP_TINY, 2)
    np.testing.assert_allclose(O.clip_normalize_pixels(T_(u8)).numpy(), px, rtol=1e-6, atol=1e-6)
