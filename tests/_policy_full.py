"""Shared by tests/test_oracle_golden.py (CPU) and tests/test_gpu_policy_sizes.py (GPU): the policy step at the sizes bench.py
times (tests/golden/inputs.py: TRAIN_FULL_CASES), its reference-chain fixture (tests/golden/train_full.npz, written by
`make_golden.py train_full` from the reference's own modules) and the oracle's every-element gradients for the same inputs."""
import functools

import numpy as np
import torch

from inputs import TRAIN_FULL_CASES, train_full_inputs
from oracle import tspo_oracle as O

TRAINED = [n for n in O.SELECTOR_KEYS if "ffn_o" not in n]
CASES = {c[0]: c for c in TRAIN_FULL_CASES}


def T_(x):
    return torch.from_numpy(np.asarray(x))


@functools.lru_cache(maxsize=None)
def oracle_prompt(name: str, b: int, noise_key: bytes):
    """The oracle's step for prompt b of case `name` under the fixture's noise: (idx [G,k], loss, {name: grad}, adv [G], scores [T]).
    T <= 1024: the reference-shaped 2.G-forward loop (O.tspo_step_autograd); longer videos: the one-forward form (2 GB instead of
    G x 2 GB of autograd state; equality of the two forms is a CPU test)."""
    _, B, T, D, H, w, tau, k, G = CASES[name]
    img, txt, clip, state, rew = train_full_inputs(name, B, T, D, G)
    noise = np.frombuffer(noise_key, np.float32).reshape(G, T)
    torch.set_num_threads(min(32, max(1, torch.get_num_threads())))
    ps = {n: T_(v) for n, v in state.items()}
    if T <= 1024:
        idx, loss, grads = O.tspo_step_autograd(ps, T_(img[b]), T_(txt[b]), T_(clip[b]), T_(noise.copy()), T_(rew[b]), k, w, tau)
        adv = O.grpo_advantage(T_(rew[b]), G)
        with torch.no_grad():
            scores, _ = O.selector_forward(ps, T_(img[b]), T_(txt[b]), T_(clip[b]), w, tau, H)
        return idx, loss, grads, adv, scores
    return O.tspo_step_autograd_one_forward(ps, T_(img[b]), T_(txt[b]), T_(clip[b]), T_(noise.copy()), T_(rew[b]), k, w, tau, H)


def oracle_mean_grads(name: str, g, prompts):
    """{param: mean over `prompts` of the oracle's per-prompt gradient} + the per-prompt (idx, loss, adv, scores)."""
    per = [oracle_prompt(name, b, np.ascontiguousarray(g[f"{name}.noise"][b]).tobytes()) for b in prompts]
    mean = {n: sum(p[2][n] for p in per) / len(per) for n in O.SELECTOR_KEYS}
    return mean, per


def check_against_fixture(name: str, g, grads: dict, what: str, rtol=5e-4, atol_rel=5e-5):
    """grads {param name: flat fp32 numpy} (the bucket of the WHOLE case: mean over all its prompts) against the reference-chain
    fixture: leading + strided samples element by element, sum / sum|.| / sum of squares / max|.| of the whole tensor."""
    for pn in TRAINED:
        got = np.asarray(grads[pn], np.float64).ravel()
        sums = g[f"{name}.gradsum.{pn}"]
        gmax = float(sums[3])
        if pn == "temporal.Self_k.bias":
            # mathematically zero (softmax is shift-invariant per query: a common key offset cancels); both sides hold rounding noise
            qb = float(g[f"{name}.gradsum.temporal.Self_q.bias"][3])
            assert np.abs(got).max() <= 1e-4 * max(qb, 1e-12), f"{what}: {pn} should vanish"
            continue
        if f"{name}.grad.{pn}" in g.files:
            ref = g[f"{name}.grad.{pn}"].astype(np.float64)
            np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_rel * gmax, err_msg=f"{what}: {pn}")
        else:
            stride = got.size // 256
            np.testing.assert_allclose(got[:256], g[f"{name}.gradsl.{pn}"], rtol=rtol, atol=atol_rel * gmax, err_msg=f"{what}: {pn} [:256]")
            np.testing.assert_allclose(got[stride // 2::stride][:256], g[f"{name}.gradst.{pn}"], rtol=rtol, atol=atol_rel * gmax,
                                       err_msg=f"{what}: {pn} strided")
        # whole-tensor checksums (sum|.| and the sum of squares are well conditioned; the plain sum is not and gets an absolute bound)
        assert abs(np.abs(got).sum() - sums[1]) <= 2e-4 * sums[1], f"{what}: {pn} sum|g| {np.abs(got).sum()} vs {sums[1]}"
        assert abs((got ** 2).sum() - sums[2]) <= 4e-4 * sums[2], f"{what}: {pn} sum g^2"
        assert abs(got.sum() - sums[0]) <= 2e-4 * sums[1], f"{what}: {pn} sum g"
        assert abs(np.abs(got).max() - gmax) <= rtol * gmax + atol_rel * gmax, f"{what}: {pn} max|g|"


def check_every_element(grads: dict, ref: dict, what: str, rtol=5e-4, atol_rel=5e-5):
    """Every element of every gradient tensor against the oracle's autograd (rtol 5e-4, atol 5e-5 x the tensor's largest
    magnitude: the criterion of test_selector_fwd_bwd_vs_oracle_autograd)."""
    worst = {}
    qb = float(ref["temporal.Self_q.bias"].abs().max())
    for pn in O.SELECTOR_KEYS:
        got = np.asarray(grads[pn], np.float32).ravel()
        if "ffn_o" in pn:
            assert np.all(got == 0), f"{what}: {pn} must stay zero (never applied, temporal_agent.py:77-79)"
            continue
        if pn == "temporal.Self_k.bias":
            assert np.abs(got).max() <= 1e-4 * max(qb, 1e-12), f"{what}: {pn} should vanish"
            continue
        r = ref[pn].numpy().ravel()
        rmax = max(float(np.abs(r).max()), 1e-12)
        worst[pn] = float(np.abs(got - r).max() / rmax)
        np.testing.assert_allclose(got, r, rtol=rtol, atol=atol_rel * rmax, err_msg=f"{what}: {pn}")
    return worst
