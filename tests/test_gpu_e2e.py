"""End-to-end pixel -> frame-index parity (north_star: "frame indices identical to the reference for greedy top-k";
reference chain model/temporal_agent.py:177-192: extract_feature -> temporal_sampling -> inference_ts).

The SAME uint8 pixels, text feature and weights go through the fp32 CPU oracle (`oracle.frames_scored_path`) and through
the HIP path (`FrameScorer`: bf16 CLIP-L/14 encode, fp32 clip score + scoring head + greedy top-k).  Unlike the piecewise
tests, the selector here sees the GPU's OWN bf16-encoded features on one side and the oracle's fp32 features on the other,
so the encode error (<= 3 % of the feature range) is carried through 1/tau = 40 into the scores.

Reported per weight set (printed; the numbers are copied into DESIGN.md section 2):
  eps       = max_t |s_hip - s_oracle|  in score units (= logits, already divided by tau), and eps * tau (cosine units)
  overlap_k = |topk_hip  ∩ topk_oracle| for k in {8, 32}
  gap_k     = oracle's k-th minus (k+1)-th largest score
Asserted:
  * the tolerance is tied to the REFERENCE'S OWN precision (it runs CLIP and the scoring head in bf16, gen_id_tspo.py:55):
    tests/golden/bf16_noise.json holds, for these very videos / weights, how far the reference's bf16 path sits from its fp32
    path (feature error, score error on six text directions, and how many of its own top-k indices move: 28 of 32 survive for an
    independent text).  HIP-vs-oracle feature error and the largest score error over the same six texts must stay within
    1.5 x those; eps * tau <= 0.02 stays as a hard ceiling;
  * every frame the oracle ranks more than 2*eps above its k-th score is selected, every frame more than 2*eps below
    it is not (what |delta| <= eps allows - checks the top-k kernel on real scores, ties included);
  * identical index lists whenever gap_k > 2*eps - the planted-needle case (k = 8) has such a gap by construction.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from inputs import (E2E_SCENARIOS, E2E_TAU, E2E_WINDOW, FULL_K, FULL_NEEDLES, FULL_T, clip_l14_state,  # noqa: E402
                    e2e_selector_state, e2e_texts, e2e_video, e2e_video_seed, full_video)
from oracle import tspo_oracle as O
from tspo_amd import ops, synth
from tspo_amd.pipeline import FrameScorer

pytestmark = pytest.mark.gpu
DEV = "cuda"
TAU, WINDOW = E2E_TAU, E2E_WINDOW
NOISE = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_noise.json")))["e2e"]


def T_(x):
    return torch.from_numpy(np.asarray(x))


def _flat(sel):
    offs = ops.flat_offsets(768)
    flat = torch.zeros(offs["__total__"][0])
    for name, (off, shape) in offs.items():
        if not name.startswith("__"):
            flat[off:off + int(np.prod(shape))] = T_(sel[name]).flatten()
    return flat.to(DEV)


HEAVY = ["heavy_tailed"] + [f"heavy_tailed_s{i}" for i in range(2, 9)]      # eight videos since round 6 (three in round 5)
_EPS, _EXTREMES = {}, {}      # scenario -> (HIP, reference-bf16) largest score error in logits; scenario -> (max-error ratio, largest-1-cos ratio)
_RATIOS = {}      # scenario -> (largest HIP score error over the six texts) / (the reference's own bf16 score error on the same video)


def test_pixels_to_indices_normal_weights():
    _pixels_to_indices("normal")


def test_pixels_to_indices_heavy_tailed_eight_videos():
    """Heavy-tailed weights (outlier channels, LayerNorm gains up to 8, like trained CLIP checkpoints) on EIGHT videos (three in round
    5).  The largest score error of ONE 64-frame video over six texts is an extreme statistic: the reference's OWN bf16 path draws
    0.37 ... 1.05 logits on these videos, the HIP path 0.36 ... 0.89, and their per-video ratio spreads over 0.61 ... 1.49 although the
    two distributions coincide (means 0.635 vs 0.659 logits, measured on MI355X, profiles/r6_f_e2e_eight_heavy_tailed_videos.txt).
    So the bounds that bind are over the SAMPLE: every video within 1.5 x the reference's own noise on that video (kept from
    round 4), the MEDIAN ratio <= 1.15, and the MEAN HIP error <= 1.1 x the MEAN reference error.  Same for the feature extremes
    (largest error, smallest cosine of 64 frames: per video <= 2 x - measured spread 0.53 ... 1.63 -, median over the videos <= 1.25 x)."""
    import statistics
    for sc in HEAVY:
        _pixels_to_indices(sc)
    r = sorted(_RATIOS[sc] for sc in HEAVY)
    med = statistics.median(r)
    hip, ref = [_EPS[sc][0] for sc in HEAVY], [_EPS[sc][1] for sc in HEAVY]
    print(f"\n[e2e heavy-tailed] HIP / reference-bf16 score-error ratios over the {len(HEAVY)} videos: {[round(x, 3) for x in r]}, median {med:.3f}; "
          f"mean error HIP {sum(hip) / len(hip):.4f} vs reference-bf16 {sum(ref) / len(ref):.4f} logits (x{sum(hip) / sum(ref):.3f})")
    assert max(r) <= 1.5 and med <= 1.15, f"per-video ratios {r}"
    assert sum(hip) <= 1.1 * sum(ref), f"mean score error {sum(hip) / len(hip)} vs the reference's own {sum(ref) / len(ref)}"
    ex_err = sorted(_EXTREMES[sc][0] for sc in HEAVY)
    ex_cos = sorted(_EXTREMES[sc][1] for sc in HEAVY)
    print(f"    feature extremes, HIP / reference-bf16 per video: max error {[round(x, 2) for x in ex_err]} (median {statistics.median(ex_err):.2f}), "
          f"largest 1-cos {[round(x, 2) for x in ex_cos]} (median {statistics.median(ex_cos):.2f})")
    assert statistics.median(ex_err) <= 1.25 and statistics.median(ex_cos) <= 1.25


def _pixels_to_indices(scenario):
    torch.set_num_threads(min(32, torch.get_num_threads()))
    cfg = synth.CLIP_L14
    weights, n, needles = E2E_SCENARIOS[scenario]
    noise = NOISE[scenario]
    assert noise["frames"] == n
    state = clip_l14_state(weights)
    sel = e2e_selector_state()
    u8 = e2e_video(n, needles, e2e_video_seed(scenario))

    # ---- oracle: fp32 arithmetic on the checkpoint's bf16-rounded matrices -------------------------------------------
    wq = {k: (T_(v).to(torch.bfloat16).float() if v.ndim >= 2 and "position_embedding" not in k else T_(v)) for k, v in state.items()}
    selp = {k: T_(v) for k, v in sel.items()}
    with torch.no_grad():
        f_ref = O.clip_vit_forward(wq, O.clip_normalize_pixels(T_(u8)), num_heads=cfg["heads"], patch=cfg["patch"])
        # the "question": what distinguishes the needle scene from the average frame (random-init CLIP features share a
        # large common component - pairwise cosine >= 0.97 - so, like a real text feature, the query is NOT along it:
        # image-text cosines come out at 0.0-0.3 as they do for trained CLIP)
        texts = e2e_texts(f_ref, needles)
        txt = texts["planted"]
        clip_ref = O.clip_cosine_scores(txt, f_ref)
        s_ref, _ = O.selector_forward(selp, f_ref, txt, clip_ref, WINDOW, TAU)

    # ---- HIP: same pixels, same text, same weights ---------------------------------------------------------------------
    clipw = ops.ClipVitWeights({k: T_(v) for k, v in state.items()}, cfg, DEV)
    scorer = FrameScorer(clipw, _flat(sel), window_size=WINDOW, score_tau=TAU)
    px = T_(u8).to(DEV)[None]
    tx = txt.to(DEV)[None]
    idx32, s_hip, f_hip = scorer(px, tx, 32)
    s_hip = s_hip[0].cpu()
    s_ref = s_ref.float()
    assert torch.isfinite(s_hip).all() and s_hip.shape == s_ref.shape

    ferr = (f_hip[0].cpu() - f_ref).abs().max().item() / f_ref.abs().max().item()
    eps = (s_hip - s_ref).abs().max().item()
    spread = (s_ref.max() - s_ref.min()).item()
    print(f"\n[e2e {scenario}, {n} frames] feature err {ferr:.4f} of range; score err eps = {eps:.4f} logits "
          f"(= {eps * TAU:.5f} in cosine units); oracle score spread {spread:.2f} logits")
    assert ferr < 3e-2
    # free-standing ceiling in cosine units: 0.02 - NOT raised - except on the videos where the reference's own bf16 path itself sits
    # above it (heavy_tailed_s3: 0.0263, s4: 0.0232), where the ceiling is 1.1 x that noise
    assert eps * TAU <= max(0.02, 1.1 * noise["max_score_eps_logits"] * TAU), f"end-to-end score error {eps * TAU} cosine units"
    # ---- the binding tolerance: the reference's own bf16-vs-fp32 noise on this video (tests/golden/bf16_noise.json) ----
    cos_all = torch.nn.functional.cosine_similarity(f_hip[0].cpu().double(), f_ref.double(), dim=-1)
    cosf = cos_all.min().item()
    rf = noise["features"]
    rms = ((f_hip[0].cpu().double() - f_ref.double()).pow(2).mean().sqrt() / f_ref.abs().max()).item()
    m1c = (1 - cos_all).mean().item()
    print(f"    features: HIP rms {rms:.5f} of range vs reference-bf16 {rf['rms_err_over_range']:.5f} (x{rms / rf['rms_err_over_range']:.2f}); mean 1-cos {m1c:.2e} vs "
          f"{rf['mean_one_minus_cos']:.2e}; max {ferr:.4f} vs {rf['err_over_range']:.4f}; largest 1-cos {1 - cosf:.2e} vs {1 - rf['min_cos']:.2e}")
    # tight on the stable statistics, loose on the extremes of a few dozen frames (they move by +-40 % between numerically
    # equivalent kernels - round 5, tests/test_gpu_ops.py:_assert_within_reference_noise)
    assert rms <= 1.25 * rf["rms_err_over_range"] and m1c <= 1.25 * rf["mean_one_minus_cos"]
    assert ferr <= 2.0 * rf["err_over_range"] and 1 - cosf <= 2.0 * (1 - rf["min_cos"])
    _EXTREMES[scenario] = (ferr / rf["err_over_range"], (1 - cosf) / (1 - rf["min_cos"]))
    eps_by_text = {}
    for tn, tq in texts.items():
        with torch.no_grad():
            sr, _ = O.selector_forward(selp, f_ref, tq, O.clip_cosine_scores(tq, f_ref), WINDOW, TAU)
        i32, sh, _ = scorer(px, tq.to(DEV)[None], 32)
        sh = sh[0].cpu()
        e = (sh - sr.float()).abs().max().item()
        eps_by_text[tn] = e
        # where the error comes from (round 5): the cosine clip score alone, the rest of the score (scoring head), and the
        # feature error projected on the text direction - next to the same three numbers of the reference's own bf16 path
        c_ref = O.clip_cosine_scores(tq, f_ref)
        c_hip = O.clip_cosine_scores(tq, f_hip[0].cpu())
        clip_e = (c_hip - c_ref).abs().max().item() / TAU
        head_e = ((sh - c_hip / TAU) - (sr.float() - c_ref / TAU)).abs().max().item()
        th = torch.nn.functional.normalize(tq.float(), dim=-1)[0]
        proj = (((f_hip[0].cpu() - f_ref) @ th) / f_ref.norm(dim=-1)).abs().max().item()
        want32 = O.topk_sorted(sr.float(), 32).tolist()
        ov = len(set(i32[0].cpu().tolist()) & set(want32))
        rt = noise["texts"][tn]
        print(f"    text {tn:14s}: eps HIP {e:.4f} logits vs reference-bf16 {rt['score_eps_logits']:.4f}; top-32 overlap with fp32: "
              f"HIP {ov}/32, reference-bf16 {rt['top32_overlap_bf16_vs_fp32']}/32")
        if "clip_eps_logits" in rt:
            print(f"         {'':14s}  clip part {clip_e:.4f} (ref {rt['clip_eps_logits']:.4f}) | head part {head_e:.4f} (ref {rt['head_eps_logits']:.4f}) | "
                  f"|d.t|/|f| {proj:.5f} (ref {rt['proj_err_max']:.5f})")
        if not tn.startswith("planted"):
            # an INDEPENDENT text (not derived from the oracle's own features): the band rule with its own eps
            order_t = torch.argsort(sr.float(), descending=True, stable=True)
            thr_t = sr.float()[order_t[31]].item()
            got = set(i32[0].cpu().tolist())
            assert set(torch.nonzero(sr.float() > thr_t + 2 * e).flatten().tolist()) <= got
            assert not (set(torch.nonzero(sr.float() < thr_t - 2 * e).flatten().tolist()) & got)
            assert ov >= rt["top32_overlap_bf16_vs_fp32"] - 4          # no worse than the reference's own bf16 path moves indices
    worst, ref_worst = max(eps_by_text.values()), noise["max_score_eps_logits"]
    print(f"    largest score error over the {len(texts)} texts: HIP {worst:.4f} logits, reference-bf16 {ref_worst:.4f} (x{worst / ref_worst:.2f})")
    assert worst <= 1.5 * ref_worst, f"score error {worst} logits > 1.5 x the reference's own bf16 noise {ref_worst}"
    _RATIOS[scenario] = worst / ref_worst
    _EPS[scenario] = (worst, ref_worst)

    order = torch.argsort(s_ref, descending=True, stable=True)
    for k in (len(needles), 8, 32):
        got = ops.topk_sorted(s_hip.to(DEV), k).cpu() if k != 32 else idx32[0].cpu()
        assert torch.equal(got, scorer(px, tx, k)[0][0].cpu())                   # the pipeline call returns the same list
        want = O.topk_sorted(s_ref, k)
        thr = s_ref[order[k - 1]].item()
        gap = thr - s_ref[order[k]].item()
        overlap = len(set(got.tolist()) & set(want.tolist()))
        must = set(torch.nonzero(s_ref > thr + 2 * eps).flatten().tolist())
        must_not = set(torch.nonzero(s_ref < thr - 2 * eps).flatten().tolist())
        print(f"    k={k:2d}: overlap {overlap}/{k}, oracle gap k/(k+1) = {gap:.4f} logits, "
              f"{len(must)} frames decided in, {n - len(must) - len(must_not)} inside the 2*eps band")
        assert must <= set(got.tolist()) and not (must_not & set(got.tolist()))
        assert got.tolist() == sorted(got.tolist()) and len(set(got.tolist())) == k
        if gap > 2 * eps:
            assert got.tolist() == want.tolist(), f"k={k}: gap {gap} > 2*eps {2 * eps} but indices differ"
    if weights != "normal":
        # heavy-tailed weights: a few outlier channels dominate the differences between frames, the planted scene does not
        # stand out from other shots even in the fp32 oracle (its top-4 holds an unrelated frame) - the band rule above is
        # the assertion; the identical-index case is the N(0, sigma) run
        return
    # the planted scene is found by both paths, with a gap the encode error cannot bridge
    kn = len(needles)
    assert O.topk_sorted(s_ref, kn).tolist() == sorted(needles)
    assert ops.topk_sorted(s_hip.to(DEV), kn).cpu().tolist() == sorted(needles)
    gap_n = (s_ref[order[kn - 1]] - s_ref[order[kn]]).item()
    assert gap_n > 2 * eps, f"needle gap {gap_n} vs 2*eps {2 * eps}: the identical-indices assertion above did not bind"


def test_full_size_pixels_to_indices_against_the_reference_fixture():
    """BASELINE configs[1] at FULL size, pixels -> indices, against outputs of the REFERENCE'S OWN chain (VERDICT r4 missing #4;
    reference: model/temporal_agent.py:177-192 extract_feature -> temporal_sampling -> inference_ts, bf16 production dtype
    mp_tools/vlmeval/vlm/gen_id_tspo.py:55).  tests/golden/full1024.npz (make_golden.py full1024, ~25 min of CPU in the authoring
    container) holds, for the 1024-frame video of tests/golden/inputs.py:full_video and two text features (the planted-scene query -
    stored, it is built from the fp32 features - and an independent N(0,1) one): the fp32 scores [1024] of installed-transformers
    CLIP-L/14 + imported MultiModal_Align, TSPOModel.inference_ts top-10 / top-32 / top-64 / bin-max-32 of them, and the same chain
    in bf16 (the reference's own noise at this size).  Here the SAME 1024 uint8 frames go through FrameScorer (one 1024-frame
    encode: every GEMM at M = 263 168 rows, remainder phase included).  Asserted:
      * features of the stored rows and all row norms within 1.5 x the reference's own bf16 feature error;
      * score error eps = max_t |s_hip - s_fixture| <= 1.5 x the reference's bf16 score error for that text;
      * band rule for top-10 / top-32 / top-64: every frame the fixture ranks more than 2 eps above (below) its k-th score is (is
        not) selected; identical lists whenever the k / (k+1) gap exceeds 2 eps - which it does for the planted scene (k = 10);
      * bin-max-32: in every bin whose fixture winner leads the bin's runner-up by more than 2 eps the same frame is returned;
      * HIP keeps at least as many of the fp32 top-32 as the reference's own bf16 path minus 4."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full1024.npz"))
    u8 = full_video()
    assert u8.shape == (FULL_T, 3, 224, 224)
    assert [int(u8.astype(np.uint64).sum()), int(u8[::97].astype(np.uint64).sum())] == z["u8.checksum"].tolist(), "not the fixture's video"
    clipw = ops.ClipVitWeights({k: T_(v) for k, v in clip_l14_state("normal").items()}, synth.CLIP_L14, DEV)
    scorer = FrameScorer(clipw, _flat(e2e_selector_state()), window_size=WINDOW, score_tau=TAU)
    px = T_(u8).to(DEV)[None]
    feats = scorer.encode(px)
    f = feats[0].float().cpu()
    rng = float(z["feat.range"])
    ferr_rows = (f[z["feat.rows"].tolist()] - T_(z["feat.values"])).abs().max().item() / rng
    nerr = (f.norm(dim=-1) - T_(z["feat.row_norms"])).abs().max().item() / rng
    ref_ferr = float(z["feat.err_over_range_bf16"])
    print(f"\n[full size, {FULL_T} frames] feature error (6 stored rows) {ferr_rows:.4f} of range, row-norm error {nerr:.4f}; reference-bf16 {ref_ferr:.4f}")
    assert ferr_rows <= 1.5 * ref_ferr and nerr <= 1.5 * ref_ferr
    for tn in ("planted", "independent"):
        txt = T_(z[f"{tn}.text"]).to(DEV)[None]
        s_fix = T_(z[f"{tn}.scores"]).float()
        ref_eps = (T_(z[f"{tn}.scores_bf16"]).float() - s_fix).abs().max().item()
        sc, _ = scorer.score(feats, txt)
        s_hip = sc[0].float().cpu()
        eps = (s_hip - s_fix).abs().max().item()
        order = torch.argsort(s_fix, descending=True, stable=True)
        print(f"    text {tn:11s}: eps HIP {eps:.4f} logits vs reference-bf16 {ref_eps:.4f} (x{eps / ref_eps:.2f}); score spread {float(s_fix.max() - s_fix.min()):.2f}")
        assert eps <= 1.5 * ref_eps, (tn, eps, ref_eps)
        for k in (len(FULL_NEEDLES), FULL_K, 64):
            want = z[f"{tn}.topk{k}"].tolist()
            assert want == O.topk_sorted(s_fix, k).tolist()                      # the oracle's rule == the reference's on the fixture's scores
            got = ops.topk_sorted(sc, k)[0].cpu().tolist()
            thr = s_fix[order[k - 1]].item()
            gap = thr - s_fix[order[k]].item()
            must = set(torch.nonzero(s_fix > thr + 2 * eps).flatten().tolist())
            must_not = set(torch.nonzero(s_fix < thr - 2 * eps).flatten().tolist())
            ov, ov_ref = len(set(got) & set(want)), len(set(z[f"{tn}.topk{k}_bf16"].tolist()) & set(want))
            print(f"      k={k:2d}: HIP keeps {ov}/{k} of the fp32 list (reference-bf16 keeps {ov_ref}/{k}); gap k/(k+1) {gap:.4f} logits, "
                  f"{len(must)} decided in, {FULL_T - len(must) - len(must_not)} inside the 2*eps band")
            assert got == sorted(got) and len(set(got)) == k
            assert must <= set(got) and not (must_not & set(got))
            assert ov >= ov_ref - 4
            if gap > 2 * eps:
                assert got == want, f"{tn} k={k}: gap {gap} > 2*eps {2 * eps} but the index lists differ"
        # bin-max (VideoMME's method, gen_id_tspo.py:83): per-bin arg-max, anchors from generate_uniform_integers
        want_b = z[f"{tn}.binmax{FULL_K}"].tolist()
        assert want_b == O.binmax(s_fix, FULL_K).tolist()
        got_b = ops.binmax(sc, FULL_K)[0].cpu().tolist()
        assert len(got_b) == len(want_b) == FULL_K and got_b == sorted(got_b)
        bounds = [-1] + [(a + b) // 2 for a, b in zip(want_b[:-1], want_b[1:])]          # only to FIND each winner's runner-up: bins are contiguous
        anchors = O.generate_uniform_integers(FULL_T - 1, FULL_K)
        slot = torch.tensor([min(range(FULL_K), key=lambda j: (abs(anchors[j] - t), j)) for t in range(FULL_T)])
        decided = same = 0
        for j in range(FULL_K):
            members = torch.nonzero(slot == j).flatten()
            vals = s_fix[members]
            top2 = torch.topk(vals, min(2, len(vals))).values
            lead = (top2[0] - top2[1]).item() if len(vals) > 1 else float("inf")
            if lead > 2 * eps:
                decided += 1
                assert got_b[j] == want_b[j], f"{tn} bin {j}: lead {lead} > 2*eps but {got_b[j]} != {want_b[j]}"
            same += int(got_b[j] == want_b[j])
        ov_ref = sum(int(a == b) for a, b in zip(z[f"{tn}.binmax{FULL_K}_bf16"].tolist(), want_b))
        print(f"      bin-max {FULL_K}: {same}/{FULL_K} bins identical (reference-bf16: {ov_ref}/{FULL_K}); {decided} bins decided by more than 2*eps")
        assert same >= ov_ref - 4
    # the planted scene is found, identically, by the fixture (the reference's fp32 AND bf16 chains) and by the HIP path
    assert z[f"planted.topk{len(FULL_NEEDLES)}"].tolist() == sorted(FULL_NEEDLES) == z[f"planted.topk{len(FULL_NEEDLES)}_bf16"].tolist()
    got = ops.topk_sorted(scorer.score(feats, T_(z["planted.text"]).to(DEV)[None])[0], len(FULL_NEEDLES))[0].cpu().tolist()
    assert got == sorted(FULL_NEEDLES)
