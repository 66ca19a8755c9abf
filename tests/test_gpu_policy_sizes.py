"""The policy step at the sizes bench.py TIMES, against the oracle and the reference-chain fixture.

`rollouts_per_s` is measured at B=4, T=512, D=768, G=8, k=16 (BT = 2048: the 64x96 / 128x128 split-7 tilings, dgrad_wgrad_kernel,
gemm_f32_tn_lds_kernel, reduce_segments), `rollouts_dp_path` on the coalesced 1 x 2 window (BT = 1024: 32x96 / 32x32 forward tiles,
split-3 backward) and the stress line at B=1, T=4096, G=16.  DESIGN 4.5 / 4.7 re-choose the tile forms by BT, so a form can be
wrong ONLY at the timed size: every gradient tensor, the gradient norm and the weights after clip + AdamW are compared here
* with the CPU oracle's autograd (oracle/tspo_oracle.py: tspo_step_autograd, the reference-shaped 2.G-forward loop of
  tspo_trainer.py:500-609; T = 4096: its one-forward form) - EVERY element, rtol 5e-4 / atol 5e-5 x the tensor's maximum, the
  criterion of test_selector_fwd_bwd_vs_oracle_autograd;
* with tests/golden/train_full.npz, written by `make_golden.py train_full` from the REFERENCE'S OWN modules (imported
  MultiModal_Align + gumbel_softmax, the literal trainer expressions): indices bit-exact under the reference's own Gumbel draws,
  advantages, losses, sampled + strided gradient elements, whole-tensor checksums, clip norm, parameters after AdamW.
"""
import numpy as np
import pytest
import torch

import _policy_full as P
from oracle import tspo_oracle as O
from tspo_amd import ops
from tspo_amd.pipeline import PolicyTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda"


def G_(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _flat(state, D):
    offs = ops.flat_offsets(D)
    flat = torch.zeros(offs["__total__"][0], dtype=torch.float32)
    for name, (off, shape) in offs.items():
        if not name.startswith("__"):
            flat[off:off + int(np.prod(shape))] = P.T_(state[name]).flatten()
    return flat.to(DEV)


def _bucket(grad, D):
    offs = ops.flat_offsets(D)
    g = grad.detach().cpu().numpy()
    return {pn: g[offs[pn][0]:offs[pn][0] + int(np.prod(offs[pn][1]))].copy() for pn in O.SELECTOR_KEYS}


def _check_update(name, g, trainer, flat0, bucket, D):
    """clip norm + the parameters after ONE AdamW step (lr 5e-4, HF defaults) against the fixture."""
    tn = float(g[f"{name}.gradnorm"])
    offs = ops.flat_offsets(D)
    clip = min(1.0, 1.0 / (tn + 1e-6))
    for pn in P.TRAINED:
        if pn == "temporal.Self_k.bias":
            continue     # its gradient is mathematically zero: BOTH sides step by +-lr on the sign of rounding noise (Adam's g / |g|)
        off = offs[pn][0]
        ref = g[f"{name}.after.{pn}"]
        ok = np.abs(bucket[pn][:ref.size]) * clip > 2e-6     # (first Adam step = lr.g / (|g| + 1e-8): ill-conditioned only where |g| ~ eps)
        got = trainer.flat[off:off + ref.size].cpu().numpy()
        assert ok.sum() > ref.size // 2, pn
        np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-4, atol=2e-6, err_msg=f"{name}: {pn} after AdamW")
        assert not np.array_equal(got, flat0[off:off + ref.size].cpu().numpy()), f"{pn} was not updated"
    o, _ = offs["temporal.ffn_o.weight"]
    assert torch.equal(trainer.flat[o:o + D * D], flat0[o:o + D * D]), "ffn_o must never be updated (SURVEY a13)"


@pytest.mark.parametrize("name", ["c2", "c4"], ids=["configs2_B4_T512_G8_k16", "configs4_B1_T4096_G16_k16"])
def test_policy_step_at_the_timed_size_every_gradient(golden, name):
    """PolicyTrainer.step - exactly what bench.py times for `rollouts_per_s` (c2) and the stress line (c4)."""
    _, B, T, D, H, w, tau, k, G = P.CASES[name]
    g = golden["train_full"]
    img, txt, clip, state, rew = P.train_full_inputs(name, B, T, D, G)
    flat0 = _flat(state, D)
    tr = PolicyTrainer(flat0.clone(), dim=D, heads=H, window_size=w, lr=5e-4, max_grad_norm=1.0)
    st = tr.step(G_(img), G_(txt), G_(clip), lambda idx: G_(rew), G, k, tau, noise=G_(g[f"{name}.noise"]))
    torch.cuda.synchronize()
    # rollouts: the reference's own draws -> the reference's own indices, bit for bit
    np.testing.assert_array_equal(st["idx"].cpu().numpy(), g[f"{name}.idx"])
    np.testing.assert_allclose(st["advantages"].cpu().numpy(), g[f"{name}.adv"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st["loss"].cpu().numpy(), g[f"{name}.loss"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(st["scores"].cpu().numpy(), g[f"{name}.scores"], rtol=2e-5, atol=2e-5 / tau)
    bucket = _bucket(tr.grad, D)
    # (1) every element of every gradient tensor against the oracle's autograd (mean over the prompts: PolicyTrainer's 1/B)
    mean, per = P.oracle_mean_grads(name, g, range(B))
    for b, (idx, *_rest) in enumerate(per):
        np.testing.assert_array_equal(idx.numpy(), g[f"{name}.idx"][b])
    worst = P.check_every_element(bucket, mean, f"{name} vs oracle autograd")
    print(f"\n[{name}: B={B} T={T} G={G}] largest |HIP - oracle| / max|oracle| per tensor: " +
          ", ".join(f"{pn.replace('temporal.', '')} {v:.1e}" for pn, v in worst.items()))
    # (2) the reference-chain fixture
    P.check_against_fixture(name, g, bucket, f"{name} vs reference fixture")
    tn = float(g[f"{name}.gradnorm"])
    assert abs(st["grad_norm_scale"][0].item() - tn) < 5e-4 * tn, (st["grad_norm_scale"], tn)
    _check_update(name, g, tr, flat0, bucket, D)


def test_coalesced_window_at_the_reference_configuration_vs_oracle(golden):
    """`rollouts_dp_path`: the reference's per_device_train_batch_size 1 x gradient_accumulation_steps 2 (train_deepspeed.sh:30-31)
    as ONE stacked rollout / backward (micro_steps=2; BT = 1024 forms) AND as two sequential micro-steps (BT = 512 small-M forms,
    TSPO_SEL_ACCUMULATE for the second) - both against the oracle's mean of the two prompts' gradients, every element, and
    clip + AdamW against the oracle's AdamW."""
    name = "c2"
    _, B, T, D, H, w, tau, k, G = P.CASES[name]
    g = golden["train_full"]
    img, txt, clip, state, rew = P.train_full_inputs(name, B, T, D, G)
    flat0 = _flat(state, D)
    for pair in ((0, 1), (2, 3)):
        sl = list(pair)
        mean, per = P.oracle_mean_grads(name, g, pair)
        f, t, c, r, nz = G_(img[sl]), G_(txt[sl]), G_(clip[sl]), G_(rew[sl]), G_(g["c2.noise"][sl])
        coa = PolicyTrainer(flat0.clone(), dim=D, heads=H, window_size=w, lr=5e-4, grad_accum_steps=2)
        st = coa.step(f, t, c, lambda idx: r, G, k, tau, noise=nz, micro_steps=2)
        np.testing.assert_array_equal(st["idx"].cpu().numpy(), g["c2.idx"][sl])
        assert "grad_norm_scale" in st, "the stacked window must end in the optimizer step"
        bc = _bucket(coa.grad, D)
        wc = P.check_every_element(bc, mean, f"coalesced 1x2 {pair} vs oracle")
        seq = PolicyTrainer(flat0.clone(), dim=D, heads=H, window_size=w, lr=5e-4, grad_accum_steps=2)
        for j in range(2):
            s1 = seq.step(f[j:j + 1], t[j:j + 1], c[j:j + 1], lambda idx, j=j: r[j:j + 1], G, k, tau, noise=nz[j:j + 1])
            np.testing.assert_array_equal(s1["idx"][0].cpu().numpy(), g["c2.idx"][sl[j]])
        assert "grad_norm_scale" in s1
        bs = _bucket(seq.grad, D)
        ws_ = P.check_every_element(bs, mean, f"sequential 1+1 {pair} vs oracle")
        print(f"\n[1x2 window, prompts {pair}] worst rel. error coalesced {max(wc.values()):.1e} / sequential {max(ws_.values()):.1e}")
        # clip + AdamW of the window against the oracle's AdamW on the oracle's gradients
        tn = torch.cat([mean[n].flatten() for n in P.TRAINED]).norm().item()
        for trn, st_ in ((coa, st), (seq, s1)):
            assert abs(st_["grad_norm_scale"][0].item() - tn) < 5e-4 * tn
        scale = O.clip_grad_scale(tn, 1.0)
        offs = ops.flat_offsets(D)
        for pn in P.TRAINED:
            if pn == "temporal.Self_k.bias":
                continue
            p0 = P.T_(state[pn])
            p1, _, _ = O.adamw_step(p0, mean[pn], torch.zeros_like(p0), torch.zeros_like(p0), 1, 5e-4, grad_scale=scale)
            ok = (mean[pn].abs() * scale > 2e-6).flatten().numpy()
            off, n = offs[pn][0], p0.numel()
            for trn in (coa, seq):
                np.testing.assert_allclose(trn.flat[off:off + n].cpu().numpy()[ok], p1.flatten().numpy()[ok], rtol=1e-4, atol=2e-6,
                                           err_msg=pn)


def test_bench_rollout_configuration_is_the_tested_one():
    """The sizes in this file ARE bench.py's: its default --rollout-cfg (configs[2]) and the stress configuration it documents."""
    import os, re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    m = re.search(r'"--rollout-cfg", default="([\d,]+)"', src)
    assert m and tuple(int(v) for v in m.group(1).split(",")) == (4, 512, 8, 16)
    assert "1,4096,16,16" in src
    assert P.CASES["c2"][1:3] + P.CASES["c2"][7:] == (4, 512, 16, 8) and P.CASES["c4"][1:3] + P.CASES["c4"][7:] == (1, 4096, 16, 16)
