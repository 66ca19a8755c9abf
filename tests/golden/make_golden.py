"""Generate the golden vectors under tests/golden/ by running THE REFERENCE.

Run in the authoring container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is executed is the reference's own code: ``model.temporal_agent``
(positional_encoding, MultiModal_Align, TSPOModel.inference_ts) and
``model.utils`` (gumbel_softmax, generate_uniform_integers, AKS_sampling)
imported from /root/reference, plus the installed ``transformers`` CLIP (the
third-party library that holds CLIP-L's arithmetic for the reference).  The
trainer lines that cannot be imported here (tspo_trainer.py:587-607 needs
trl/deepspeed) are evaluated by autograd over the imported reference modules
using the literal expressions of those lines.

Inputs are NOT stored: they are regenerated from ``tspo_amd.synth`` (pure
hash of (seed, index)), so the .npz files hold expected outputs only (plus the
Gumbel noise that was drawn).  Nothing of the reference's source is copied.
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import contextlib
import io

import numpy as np
import torch

from model.temporal_agent import MultiModal_Align, TSPOModel, positional_encoding  # reference
from model.utils import gumbel_softmax, generate_uniform_integers, AKS_sampling     # reference
from tspo_amd import synth
sys.path.insert(0, HERE)
from inputs import (SELECTOR_CASES, GUMBEL_CASES, TRAIN_CASES, TRAIN_FULL_CASES, CLIP_CASES, selector_inputs, gumbel_logits,
                    train_inputs, train_full_inputs, clip_pixels)

torch.manual_seed(0)
torch.set_num_threads(8)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ref_selector(dim, heads, state):
    m = MultiModal_Align(dim=dim, num_heads=heads).float()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return m


def gen_selector(out):
    tspo_like = TSPOModel.inference_ts  # unbound; `self` unused by the method body
    for name, T, D, H, w, tau, M, ks in SELECTOR_CASES:
        img, txt, clip, state = selector_inputs(name, T, D, M)
        m = ref_selector(D, H, state)
        with torch.no_grad():
            s, h = m(torch.from_numpy(img), torch.from_numpy(txt), torch.from_numpy(clip),
                     window_size=w, score_tau=tau)
        out[f"{name}.scores"] = s.numpy()
        if T * D <= 4096:
            out[f"{name}.attn"] = h.numpy()
        else:
            out[f"{name}.attn_rows"] = h[0, [0, 1, T // 2, T - 1]].numpy()
            out[f"{name}.attn_sum"] = np.array([h.double().sum().item(), h.double().abs().sum().item()])
        for k in ks:
            out[f"{name}.topk{k}"] = quiet(tspo_like, None, s, "topk", k)[0].numpy()
            out[f"{name}.binmax{k}"] = quiet(tspo_like, None, s, "bin-max", k)[0].numpy()
        if T >= 16:
            for k in (8, 16):
                out[f"{name}.aks{k}"] = np.asarray(quiet(AKS_sampling, s.float().numpy(), k), dtype=np.int64)
        # window mask of the reference (python double loop) for the small cases
        if T <= 64:
            out[f"{name}.mask"] = m.create_window_mask(T, window_size=w).numpy().astype(np.uint8)


def gen_misc(out):
    out["pe_4_8"] = positional_encoding(4, 8).numpy()
    out["pe_5_6"] = positional_encoding(5, 6).numpy()
    for t, l in [(127, 16), (1023, 64), (49, 50), (6, 7), (10, 1), (1023, 32), (99, 8), (510, 5)]:
        out[f"uni_{t}_{l}"] = np.asarray(generate_uniform_integers(t, l), dtype=np.int64)
    # ties: declared rule (lowest index) must at least agree with the reference's *value* multiset
    s = torch.tensor([0.5, 2.0, 2.0, -1.0, 2.0, 0.5, 2.0, 0.25])
    out["ties.scores"] = s.numpy()
    out["ties.top3_values"] = torch.topk(s, 3)[0].numpy()


def draw_gumbel(T, seed):
    """The draw F.gumbel_softmax makes internally, replayed under the same seed."""
    torch.manual_seed(seed)
    return -torch.empty(T, 1).exponential_().log()


def gen_gumbel(out):
    for name, T, k, G, scale in GUMBEL_CASES:
        logits = torch.from_numpy(gumbel_logits(T, scale))
        noise = np.zeros((G, T), np.float32)
        idxs = np.zeros((G, k), np.int64)
        probs = np.zeros((G, T), np.float32)
        for g in range(G):
            seed = 9000 + 17 * g + T
            noise[g] = draw_gumbel(T, seed)[:, 0].numpy()
            torch.manual_seed(seed)
            idx, p, lp = gumbel_softmax(logits.unsqueeze(1), sample_len=k)
            idxs[g], probs[g] = idx.numpy(), p.numpy()
        out[f"{name}.noise"] = noise
        out[f"{name}.idx"] = idxs
        out[f"{name}.probs"] = probs
        out[f"{name}.logp"] = lp.numpy()


def trainer_step(m, img, txt, clip, w, tau, k, G, rewards, seed0):
    """tspo_trainer.py:500-609 for one prompt, on the imported reference modules."""
    T = img.shape[0]
    all_ts, noise = [], np.zeros((G, T), np.float32)
    with torch.no_grad():
        for g in range(G):                                   # rollout loop (:508-537)
            conf, _ = m(img, txt, clip, w, tau)              # llava_qwen.py:133
            noise[g] = draw_gumbel(T, seed0 + g)[:, 0].numpy()
            torch.manual_seed(seed0 + g)
            idx, _, _ = gumbel_softmax(conf.unsqueeze(1), sample_len=k)   # llava_qwen.py:137
            all_ts.append((idx.clone(), idx))
    ts_logps_batch = []
    for g in range(G):                                       # re-eval loop (:540-552)
        conf, _ = m(img, txt, clip, w, tau)
        _, _, logp_ts = gumbel_softmax(conf.unsqueeze(1), sample_len=k)   # fresh (unused) noise, llava_qwen.py:140
        ts_logps_batch.append(logp_ts[all_ts[g][1]])        # :544
    mean_g = rewards.view(-1, G).mean(dim=1).repeat_interleave(G, dim=0)   # :587-592
    std_g = rewards.view(-1, G).std(dim=1).repeat_interleave(G, dim=0)
    advantages = (rewards - mean_g) / (std_g + 1e-4)
    loss_list = 0.0
    for b in range(G):                                       # :594-607
        ts_probs_item = torch.exp(ts_logps_batch[b] - ts_logps_batch[b].detach()).mean()
        loss_list = loss_list + (-(ts_probs_item * advantages[b]))
    loss = loss_list / G
    m.zero_grad()
    loss.backward()
    return all_ts, noise, advantages, loss, conf


def gen_train(out):
    for name, T, D, H, w, tau, k, G in TRAIN_CASES:
        img, txt, clip, state, rew = train_inputs(name, T, D, G)
        m = ref_selector(D, H, state)
        rew = torch.from_numpy(rew)
        all_ts, noise, adv, loss, conf = trainer_step(
            m, torch.from_numpy(img), torch.from_numpy(txt), torch.from_numpy(clip), w, tau, k, G, rew, 4242 + T)
        out[f"{name}.noise"] = noise
        out[f"{name}.rewards"] = rew.numpy()
        out[f"{name}.idx"] = np.stack([t[1].numpy() for t in all_ts])
        out[f"{name}.adv"] = adv.numpy()
        out[f"{name}.loss"] = np.array(loss.item(), np.float64)
        out[f"{name}.scores"] = conf.detach().numpy()
        # dL/dscores by autograd on a detached leaf (closed form is checked against this)
        leaf = conf.detach().clone().requires_grad_(True)
        lp = torch.softmax(leaf, dim=0).log()
        L = 0.0
        for g in range(G):
            sel = lp[all_ts[g][1]]
            L = L + (-(torch.exp(sel - sel.detach()).mean() * adv[g]))
        (L / G).backward()
        out[f"{name}.dscores"] = leaf.grad.numpy()
        for pn, p in m.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            if g.numel() <= 4096 + 64:
                out[f"{name}.grad.{pn}"] = g.numpy().copy()
            else:
                gf = g.flatten()
                out[f"{name}.gradsl.{pn}"] = gf[:256].numpy().copy()
                out[f"{name}.gradsum.{pn}"] = np.array([gf.double().sum().item(), gf.double().abs().sum().item(),
                                                         (gf.double() ** 2).sum().item()])
        # one AdamW step with HF-Trainer defaults (lr 5e-4, betas .9/.999, eps 1e-8, wd 0, clip-norm 1.0)
        params = [p for p in m.parameters() if p.grad is not None]
        tn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        out[f"{name}.gradnorm"] = np.array(tn.item(), np.float64)
        opt = torch.optim.AdamW(params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        opt.step()
        for pn, p in m.named_parameters():
            if p.grad is None:
                continue
            pf = p.detach().flatten()
            out[f"{name}.after.{pn}"] = pf[:256].numpy().copy()

    # advantage edge cases (tspo_trainer.py:587-592): all-equal rewards, general-type acc+1, G=2
    for nm, r, G in [("eq", [1.0, 1.0, 1.0, 1.0], 4), ("gen", [2.0, 1.0, 1.0, 2.0, 1.0, 1.0, 1.0, 2.0], 8),
                     ("two", [0.3, 1.7, 0.0, 0.0], 2), ("bg", list(np.linspace(0, 2, 12)), 4)]:
        rt = torch.tensor(r, dtype=torch.float32)
        mean_g = rt.view(-1, G).mean(dim=1).repeat_interleave(G, dim=0)
        std_g = rt.view(-1, G).std(dim=1).repeat_interleave(G, dim=0)
        out[f"adv.{nm}.r"] = rt.numpy()
        out[f"adv.{nm}.a"] = ((rt - mean_g) / (std_g + 1e-4)).numpy()
        out[f"adv.{nm}.G"] = np.array(G)


def trainer_step_streamed(m, img, txt, clip, w, tau, k, G, rewards, seed0):
    """trainer_step for LONG videos (T = 4096: the reference's dense T x T attention keeps ~2.5 GB of autograd state per
    re-evaluation, and tspo_trainer.py:540-552 holds all G of them): the same reference modules and the same literal expressions,
    but rollout g's term -(exp(lp - sg(lp)).mean() * A_g) / G is back-propagated as soon as it is formed - the gradient is the
    same sum (fp32 addition order of G terms aside), the memory one graph."""
    T = img.shape[0]
    all_ts, noise = [], np.zeros((G, T), np.float32)
    with torch.no_grad():
        conf, _ = m(img, txt, clip, w, tau)
        for g in range(G):
            noise[g] = draw_gumbel(T, seed0 + g)[:, 0].numpy()
            torch.manual_seed(seed0 + g)
            idx, _, _ = gumbel_softmax(conf.unsqueeze(1), sample_len=k)
            all_ts.append((idx.clone(), idx))
    mean_g = rewards.view(-1, G).mean(dim=1).repeat_interleave(G, dim=0)
    std_g = rewards.view(-1, G).std(dim=1).repeat_interleave(G, dim=0)
    advantages = (rewards - mean_g) / (std_g + 1e-4)
    m.zero_grad()
    loss = 0.0
    for g in range(G):
        conf, _ = m(img, txt, clip, w, tau)
        _, _, logp_ts = gumbel_softmax(conf.unsqueeze(1), sample_len=k)
        sel = logp_ts[all_ts[g][1]]
        term = -(torch.exp(sel - sel.detach()).mean() * advantages[g]) / G
        term.backward()
        loss = loss + term.item()
        print(f"    rollout {g + 1}/{G} re-evaluated", flush=True)
    return all_ts, noise, advantages, torch.tensor(loss), conf.detach()


def gen_train_full(out):
    """The policy step at the sizes bench.py times (BASELINE configs[2]: B=4, T=512, G=8, k=16; the policy side of configs[4]:
    B=1, T=4096, G=16, k=16), through the reference's own modules: every prompt is one tspo_trainer.py:500-609 step, the
    bucket is the MEAN of the prompts' gradients (what data-parallel averaging / PolicyTrainer's 1/B scale produce), then
    clip_grad_norm_(1.0) + one AdamW step.  Stored: the noise drawn, indices, advantages, per-prompt losses, the scores, and
    for every parameter gradient 256 leading + 256 strided elements, three checksums, the small tensors whole; the clip norm
    and 256 elements of every parameter after the update."""
    import time
    for name, B, T, D, H, w, tau, k, G in TRAIN_FULL_CASES:
        t0 = time.time()
        img, txt, clip, state, rew = train_full_inputs(name, B, T, D, G)
        m = ref_selector(D, H, state)
        acc = {pn: torch.zeros_like(p) for pn, p in m.named_parameters()}
        noises, idxs, advs, losses, scores = [], [], [], [], []
        for b in range(B):
            step = trainer_step if T <= 1024 else trainer_step_streamed
            all_ts, noise, adv, loss, conf = step(m, torch.from_numpy(img[b]), torch.from_numpy(txt[b]), torch.from_numpy(clip[b]),
                                                  w, tau, k, G, torch.from_numpy(rew[b]), 6000 + 100 * b + T)
            for pn, p in m.named_parameters():
                if p.grad is not None:
                    acc[pn] += p.grad / B
            noises.append(noise); idxs.append(np.stack([t[1].numpy() for t in all_ts])); advs.append(adv.numpy())
            losses.append(float(loss)); scores.append(conf.detach().numpy())
            print(f"  {name}: prompt {b + 1}/{B} done ({time.time() - t0:.0f}s)", flush=True)
        out[f"{name}.noise"] = np.stack(noises)
        out[f"{name}.idx"] = np.stack(idxs)
        out[f"{name}.adv"] = np.stack(advs)
        out[f"{name}.loss"] = np.array(losses, np.float64)
        out[f"{name}.scores"] = np.stack(scores)
        for pn, p in m.named_parameters():
            p.grad = acc[pn].clone()
            gf = acc[pn].flatten()
            if gf.numel() <= 4096:
                out[f"{name}.grad.{pn}"] = gf.numpy().copy()
            else:
                stride = gf.numel() // 256
                out[f"{name}.gradsl.{pn}"] = gf[:256].numpy().copy()
                out[f"{name}.gradst.{pn}"] = gf[stride // 2::stride][:256].numpy().copy()
            out[f"{name}.gradsum.{pn}"] = np.array([gf.double().sum().item(), gf.double().abs().sum().item(),
                                                     (gf.double() ** 2).sum().item(), gf.abs().max().item()])
        params = [p for pn, p in m.named_parameters() if "ffn_o" not in pn]
        tn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        out[f"{name}.gradnorm"] = np.array(tn.item(), np.float64)
        opt = torch.optim.AdamW(params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        opt.step()
        for pn, p in m.named_parameters():
            if "ffn_o" not in pn:
                out[f"{name}.after.{pn}"] = p.detach().flatten()[:256].numpy().copy()
        print(f"  {name}: |g| = {tn.item():.6f}, losses {losses}", flush=True)


def gen_clip(out):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    def run(cfgd, n_frames, tag, want_hidden):
        cfg = CLIPVisionConfig(hidden_size=cfgd["hidden"], intermediate_size=cfgd["mlp"],
                               num_hidden_layers=cfgd["layers"], num_attention_heads=cfgd["heads"],
                               image_size=cfgd["image"], patch_size=cfgd["patch"],
                               projection_dim=cfgd["proj"], hidden_act="quick_gelu", layer_norm_eps=1e-5,
                               attn_implementation="eager")
        model = CLIPVisionModelWithProjection(cfg).float().eval()
        state = synth.clip_vision_state(**cfgd)
        missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
        assert not missing.unexpected_keys, missing
        assert all("position_ids" in k for k in missing.missing_keys), missing
        px = torch.from_numpy(clip_pixels(cfgd, n_frames)[1])
        with torch.no_grad():
            o = model(pixel_values=px, output_hidden_states=want_hidden)
        out[f"{tag}.feat"] = o.image_embeds.numpy()
        if want_hidden:
            # hidden_states[0] = embeddings output BEFORE pre-LN in transformers; [i] = after layer i
            for i, h in enumerate(o.hidden_states):
                out[f"{tag}.hidden{i}"] = h.numpy()
        return model, px

    for tag, cfgd, n in CLIP_CASES:
        run(cfgd, n, tag, tag == "clip_tiny")


def _ref_functions(relpath, names, namespace):
    """Evaluate the named top-level functions of a reference file that cannot be imported here (its module imports
    trl / deepspeed / decord / math_verify at the top): the function definitions are taken from the file's AST at
    generation time and executed in `namespace`.  Nothing of the source text is stored."""
    import ast
    src = open(os.path.join("/root/reference", relpath)).read()
    tree = ast.parse(src)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), (relpath, [n.name for n in picked])
    mod = ast.Module(body=picked, type_ignores=[])
    exec(compile(mod, relpath, "exec"), namespace)
    return namespace


GLUE_STRINGS = [
    "B", "(B).", "b) because a cat appears", "The answer is C.", "Answer: (d)", "  E  ", "A.", "(a)", "a", "option b",
    "because", "", "   ", "f", "zebra", "Z) none", "I think it is (C) or (D)", "<answer>B</answer>", "<answer> c </answer>",
    "<think>it is a dog</think><answer>A</answer>", "<think>x</think>\n  <answer>D</answer> trailing",
    "<answer>A</answer><think>late</think>", "no tags at all", "12345", "a-b", "A,B", "xAy", "_b_", "b2", "3c", "E!",
    "the end", "d\ne", "Bee", "(B", "B)", "[c]", "{d}", "e.g. this", "i.e. that", "ABCD",
]


def gen_glue(_out):
    """Reward glue (src/open_tspo/tspo.py:86-166, tspo_trainer.py:570-573) and the needle-in-haystack sample builder
    (src/open_tspo/trainer/utils.py:15-25,177-200) evaluated from the reference's own function bodies.  math_verify is
    not installed here: parse() raises, so accuracy_reward takes its string-matching branch (the documented fallback of
    the reference itself, tspo.py:118-136).  Written as JSON (strings + nested lists) to tests/golden/glue.json."""
    import datetime as _dt
    import json
    import re

    def _no_math_verify(*a, **k):
        raise RuntimeError("math_verify is not available")

    ns = {"re": re, "os": os, "torch": torch, "np": np, "datetime": _dt.datetime, "parse": _no_math_verify,
          "verify": _no_math_verify}
    _ref_functions("src/open_tspo/tspo.py", ["map_prediction_to_option", "accuracy_reward", "temporal_localization_reward",
                                            "format_reward"], ns)
    g = {"strings": GLUE_STRINGS}
    g["map_prediction_to_option"] = [ns["map_prediction_to_option"](x) for x in GLUE_STRINGS]
    g["format_reward"] = ns["format_reward"]([[{"content": x}] for x in GLUE_STRINGS])
    sols = ["<answer>B</answer>", "B", "<answer>(c)</answer>", "<think>t</think><answer> D. the dog </answer>", "e", "<answer></answer>", ""]
    g["solutions"] = sols
    g["accuracy_reward"] = []
    for sol in sols:
        comps = [[{"content": x}] for x in GLUE_STRINGS]
        sel = [(None, torch.zeros(1, dtype=torch.long))] * len(comps)
        g["accuracy_reward"].append(ns["accuracy_reward"](comps, [sol] * len(comps), sel, None))
    # temporal localisation reward: sum(mask[idx]) / k for each rollout (sel_idx is the (idx.clone(), idx) pair)
    tl = []
    for T, k, G, seed in [(64, 8, 4, 1), (650, 16, 8, 2), (50, 8, 8, 3), (7, 7, 2, 4)]:
        mask = synth.uniform((T,), 600 + seed) > 0.6
        idx = [np.sort(np.argsort(synth.uniform((T,), 700 + seed * 31 + gidx))[:k]) for gidx in range(G)]
        sel = [(torch.from_numpy(i), torch.from_numpy(i)) for i in idx]
        r = ns["temporal_localization_reward"](None, None, sel, torch.from_numpy(mask))
        tl.append({"T": T, "k": k, "G": G, "seed": seed, "rewards": r})
    g["temporal_localization_reward"] = tl
    # combination of the reward columns (tspo_trainer.py:570-573), both item types
    rpf = torch.tensor([[1.0, 0.25], [0.0, 0.5], [1.0, 1.0], [0.0, 0.0]])
    g["combine"] = {"rewards_per_func": rpf.tolist(), "specific": rpf.sum(dim=1).tolist(),
                    "general": (rpf[:, 0:1].sum(dim=1) + 1).tolist()}

    # ---- needle-in-haystack builder: frames carry their own identity, so the merged "video" IS the index map ----
    ns2 = {"np": np, "torch": torch}
    _ref_functions("src/open_tspo/trainer/utils.py", ["repeat_videos", "shuffle_clips"], ns2)
    hay = []
    for L, rep, n_wrong, sample_len, seed in [(128, 4, 12, 50, 11), (128, 1, 12, 50, 12), (77, 3, 12, 50, 13),
                                              (40, 2, 12, 50, 14), (50, 4, 3, 50, 15), (300, 2, 5, 20, 16)]:
        np.random.seed(seed)
        video = np.arange(L, dtype=np.int64).reshape(L, 1, 1, 1) * np.ones((1, 1, 1, 3), np.int64)      # frame i == i
        true_videos = ns2["repeat_videos"](video, repeat_times=rep, sample_len=sample_len)
        glen = len(true_videos[0])
        wrong = [(-1 - w) * np.ones((glen, 1, 1, 3), np.int64) for w in range(n_wrong)]               # distractor w == -1-w
        merged, mask = ns2["shuffle_clips"](true_videos, wrong)
        hay.append({"L": L, "repeat_times": rep, "n_wrong": n_wrong, "sample_len": sample_len, "seed": seed,
                    "frame_ids": merged[:, 0, 0, 0].tolist(), "mask": mask.numpy().astype(int).tolist()})
    g["haystack"] = hay

    # ---- frame-index join (mp_tools/change_score_tch.py): the reference's flat script itself, run on temp files.  Inputs
    #      (annotation list, results {index: [frame numbers]}) and the JSON text it writes are recorded. ----
    import pickle
    import subprocess
    import tempfile
    joins = []
    cases = [
        ("VideoMME", "videomme", [{"question_id": "001-1", "video": "a.mp4", "duration": "short"},
                                  {"question_id": "001-2", "video": "a.mp4"}, {"question_id": "777-3", "video": "z.mp4"}],
         [["001-1", [0.0, 30.0, 61.0, 1502.0]], ["777-3", [12.0]], ["unused", [1.0]]]),
        ("MLVU", "mlvu", [{"question_id": 5, "question": "q?"}, {"question_id": 6, "question": "r?"}],
         [[6, [3.0, 4.5]], [5, []]]),
        ("LongVideoBench", "lvb_val", [{"id": "abc", "question_id": "ignored"}, {"id": "def"}, {"id": 9}],
         [["def", [100.0, 250.0]], [9, [0.0]], ["ignored", [7.0]]]),
    ]
    for data, json_name, anno, results in cases:
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, "evaluation", "jsons"))
            os.makedirs(os.path.join(td, "evaluation", "jsons_idx"))
            os.makedirs(os.path.join(td, "work_dir"))
            with open(os.path.join(td, "evaluation", "jsons", f"{json_name}.json"), "w") as f:
                json.dump(anno, f)
            with open(os.path.join(td, "work_dir", f"run7_{data}_supp.pkl"), "wb") as f:
                pickle.dump({k: v for k, v in results}, f)
            r = subprocess.run([sys.executable, "/root/reference/mp_tools/change_score_tch.py", "--base_anno_path",
                                os.path.join(td, "evaluation"), "--data", data, "--name", "run7"], cwd=td, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            out_text = open(os.path.join(td, "evaluation", "jsons_idx", f"run7_{data}_frameIdx.json")).read()
        joins.append({"data": data, "json_name": json_name, "name": "run7", "anno": anno, "results": results,
                      "written_relpath": os.path.join("jsons_idx", f"run7_{data}_frameIdx.json"), "written_text": out_text,
                      "printed_missing": r.stdout.split()})
    g["frame_idx_join"] = joins

    # ---- frame plans of the harness (lmms-eval llava_vid_tspo.py:315-380): the reference's own method bodies, taken from the
    #      class by AST, run against a fake decord reader ----
    import ast as _ast
    src = open("/root/reference/lmms-eval/lmms_eval/models/simple/llava_vid_tspo.py").read()
    cls = [n for n in _ast.parse(src).body if isinstance(n, _ast.ClassDef)]
    meths = [m for c in cls for m in c.body if isinstance(m, _ast.FunctionDef) and m.name in ("load_video", "load_video_sampled",
                                                                                             "load_video_index")]
    assert sorted(m.name for m in meths) == ["load_video", "load_video_index", "load_video_sampled"]

    class _Batch:
        def __init__(self, idx):
            self.idx = idx

        def asnumpy(self):
            return np.asarray([float(i) for i in self.idx])

    class _Reader:
        def __init__(self, path, ctx=None, num_threads=1):
            self.n, self.fps = path

        def __len__(self):
            return self.n

        def get_avg_fps(self):
            return self.fps

        def get_batch(self, idx):
            return _Batch(idx)

    ns3 = {"np": np, "VideoReader": _Reader, "cpu": lambda i: None}
    exec(compile(_ast.Module(body=meths, type_ignores=[]), "llava_vid_tspo.py", "exec"), ns3)
    plans = []
    for total, avg_fps, fps, mx in [(9000, 30.0, 1, 64), (1500, 29.97, 1, 64), (900, 25.0, 1, 64), (9000, 30.0, 2, 32),
                                    (50, 30.0, 1, 8), (100000, 23.976, 1, 64)]:
        fr, ft, vt = ns3["load_video"](None, (total, avg_fps), mx, fps)
        fr2, ft2, vt2 = ns3["load_video"](None, (total, avg_fps), mx, fps, force_sample=True)
        step = round(avg_fps / fps)
        n_cand = len(range(0, total, step))
        pick = sorted(int(v) for v in np.argsort(synth.uniform((n_cand,), 900 + total))[:mx]) if n_cand > mx else None
        agent = lambda proc, cand, problem, sample_num, window_size, method, processor_type: (torch.tensor(pick), None)
        fs, fts, vts = ns3["load_video_sampled"](None, (total, avg_fps), mx, fps, "q", agent, None)
        docs = [[float(step * v) for v in np.sort(np.argsort(synth.uniform((n_cand,), 950 + total))[:mx])][::-1],   # unsorted on purpose
                [float(step * v) for v in range(min(5, n_cand))]]                                                   # too short -> uniform
        idx_plans = []
        for d in docs:
            fi, fti, vti = ns3["load_video_index"](None, (total, avg_fps), mx, fps, {"frame_idx": d})
            idx_plans.append({"doc_frame_idx": d, "frames": fi.tolist(), "frame_time": fti, "video_time": vti})
        plans.append({"total": total, "avg_fps": avg_fps, "fps": fps, "max_frames_num": mx,
                      "uniform": {"frames": fr.tolist(), "frame_time": ft, "video_time": vt},
                      "uniform_forced": {"frames": fr2.tolist(), "frame_time": ft2, "video_time": vt2},
                      "sampled": {"pick": pick, "frames": fs.tolist(), "frame_time": fts, "video_time": vts},
                      "from_index": idx_plans})
    g["frame_plans"] = plans

    # ---- the partition helper of adaptive keyframe sampling (model/utils.py:83-126), its own function body ----
    import heapq as _heapq
    ns4 = _ref_functions("model/utils.py", ["meanstd"], {"np": np, "heapq": _heapq})
    ms = []
    for T, n, t1, depth, seed in [(64, 8, 0.2, 3, 11), (300, 16, 0.2, 3, 12), (97, 8, 0.05, 4, 13), (40, 8, 0.5, 2, 14), (128, 32, 0.2, 5, 15)]:
        sc = synth.uniform((T,), 700 + seed).astype(np.float64)
        sc[::7] = sc[0]                                               # ties
        z = (sc - sc.min()) / (sc.max() - sc.min())
        a, b = ns4["meanstd"](T, [dict(score=z, depth=0)], n, [list(range(T))], t1, -100, depth)
        ms.append({"T": T, "n": n, "t1": t1, "all_depth": depth, "seed": 700 + seed, "depths": [x["depth"] for x in a],
                   "frames": [[int(v) for v in f] for f in b]})
    g["meanstd"] = ms
    path = os.path.join(HERE, "glue.json")
    with open(path, "w") as f:
        json.dump(g, f, indent=0)
    print(f"glue: -> {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


PUBLISHED = [("LongVideoBench", "id", "lvb_val"), ("MLVU", "question_id", "mlvu"), ("VideoMME", "question_id", "videomme")]


def gen_published(_out):
    """The reference-held published outputs at the frame-index boundary (evaluation/jsons_idx/TSPO_*_frameIdx.json, read by
    lmms-eval llava_vid_tspo.py:362-380; written by mp_tools/change_score_tch.py:22-44 from evaluation/jsons/*.json): per-file
    invariants plus a deterministic sample of whole docs -> tests/golden/published.json.  They cannot be REPRODUCED offline (no
    weights, no videos); they pin the formats: join keys, float frame numbers, ascending order, the short-video branch."""
    import hashlib
    import json
    import math
    g = {}
    for ds, key, anno_name in PUBLISHED:
        path = f"/root/reference/evaluation/jsons_idx/TSPO_{ds}_frameIdx.json"
        text = open(path).read()
        docs = json.loads(text)
        anno = json.load(open(f"/root/reference/evaluation/jsons/{anno_name}.json"))
        assert text == json.dumps(docs) and len(anno) == len(docs)
        assert all(list(d)[-1] == "frame_idx" for d in docs)                 # the join appends the field (change_score_tch.py:42)
        assert all({k: v for k, v in d.items() if k != "frame_idx"} == a for d, a in zip(docs, anno))
        lens = [len(d["frame_idx"]) for d in docs]

        def step_of(f):
            s = 0
            for v in f:
                s = math.gcd(s, int(v))
            return s
        short = [i for i, d in enumerate(docs) if len(d["frame_idx"]) < 64]
        slow = [i for i, d in enumerate(docs) if len(d["frame_idx"]) == 64 and step_of(d["frame_idx"]) == 1]
        pick = sorted(set([0, 1] + [len(docs) * j // 7 for j in range(1, 7)] + short[:3] + slow[:2]))
        g[ds] = {
            "join_key": key, "anno_file": anno_name, "n_docs": len(docs), "sha256": hashlib.sha256(text.encode()).hexdigest(),
            "bytes": len(text), "distinct_keys": len({d[key] for d in docs}),
            "n_with_64": sum(n == 64 for n in lens), "n_short": len(short), "max_len": max(lens), "min_len": min(lens),
            "all_ascending": all(d["frame_idx"] == sorted(d["frame_idx"]) for d in docs),
            "all_distinct": all(len(set(d["frame_idx"])) == len(d["frame_idx"]) for d in docs),
            "all_float_integers": all(isinstance(v, float) and v.is_integer() for d in docs for v in d["frame_idx"]),
            # a doc shorter than 64 entries lists EVERY 1-fps candidate 0, s, 2s, ... (gen_id_tspo.py:83-92: no selection when T <= 64)
            "short_docs_are_all_candidates": all(
                [int(v) for v in docs[i]["frame_idx"]] == [j * int(docs[i]["frame_idx"][1] - docs[i]["frame_idx"][0])
                                                           for j in range(len(docs[i]["frame_idx"]))] for i in short),
            "sample_positions": pick, "sample_docs": [docs[i] for i in pick],
        }
    path = os.path.join(HERE, "published.json")
    with open(path, "w") as f:
        json.dump(g, f, indent=0)
    print(f"published: -> {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---- helpers of the `noise` and `full1024` groups: the reference's arithmetic in fp32 and in bf16 (its production dtype) ----
def _clip_model(cfgd, state):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    cfg = CLIPVisionConfig(hidden_size=cfgd["hidden"], intermediate_size=cfgd["mlp"], num_hidden_layers=cfgd["layers"],
                           num_attention_heads=cfgd["heads"], image_size=cfgd["image"], patch_size=cfgd["patch"],
                           projection_dim=cfgd["proj"], hidden_act="quick_gelu", layer_norm_eps=1e-5, attn_implementation="eager")
    model = CLIPVisionModelWithProjection(cfg).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
    return model.to(torch.bfloat16)          # parameters rounded once; .float() below keeps the rounded values


def _both(model, px, progress=None):
    out32, out16 = [], []
    with torch.no_grad():
        for i in range(0, px.shape[0], 16):
            out16.append(model(pixel_values=px[i:i + 16].to(torch.bfloat16)).image_embeds.float())
            if progress and i % 128 == 0:
                print(f"   {progress}: bf16 {i}/{px.shape[0]}", flush=True)
        model.float()
        for i in range(0, px.shape[0], 16):
            out32.append(model(pixel_values=px[i:i + 16]).image_embeds)
            if progress and i % 128 == 0:
                print(f"   {progress}: fp32 {i}/{px.shape[0]}", flush=True)
    return torch.cat(out32), torch.cat(out16)


def _feat_stats(f32, f16):
    cos = torch.nn.functional.cosine_similarity(f32, f16, dim=-1)
    rng = float(f32.abs().max())
    # max error / smallest cosine are EXTREME statistics of a few dozen frames (they move by +-40 % between two numerically
    # equivalent kernels - round 5: the same attention with a different rescaling point); the RMS error and the mean 1 - cos are
    # the stable ones, and what the tests bound tightly
    return {"err_over_range": float((f16 - f32).abs().max() / rng), "min_cos": float(cos.min()), "range": rng,
            "rms_err_over_range": float((f16 - f32).double().pow(2).mean().sqrt() / rng),
            "mean_one_minus_cos": float((1.0 - cos.double()).mean())}


def _normalize_u8(u8):
    from transformers.image_utils import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
    m = torch.tensor(OPENAI_CLIP_MEAN).view(1, 3, 1, 1)
    sd = torch.tensor(OPENAI_CLIP_STD).view(1, 3, 1, 1)
    return (torch.from_numpy(u8).float() / 255.0 - m) / sd


def _head_both(f32, f16, txt, sel, window, tau):
    """scores of the reference's scoring head: fp32 on the fp32 features, and bf16 module on the bf16 features"""
    cs = torch.nn.CosineSimilarity(dim=-1)
    with torch.no_grad():
        c32 = cs(txt, f32)
        s32, _ = ref_selector(768, 8, sel)(f32, txt, c32, window_size=window, score_tau=tau)
        m16 = ref_selector(768, 8, sel).to(torch.bfloat16)
        f16b, t16 = f16.to(torch.bfloat16), txt.to(torch.bfloat16)
        c16 = cs(t16, f16b)
        s16, _ = m16(f16b, t16, c16, window_size=window, score_tau=tau)
    return s32, s16.float(), c32, c16.float()


def _head_noise(f32, f16, txt, sel, window, tau):
    tspo_like = TSPOModel.inference_ts
    s32, s16, c32, c16 = _head_both(f32, f16, txt, sel, window, tau)
    eps = float((s16 - s32).abs().max())
    r = {"score_eps_logits": eps, "score_eps_cosine_units": eps * tau, "score_spread_logits": float(s32.max() - s32.min())}
    for k in (8, 32):
        a = quiet(tspo_like, None, s32, "topk", k)[0].tolist()
        b = quiet(tspo_like, None, s16, "topk", k)[0].tolist()
        r[f"top{k}_overlap_bf16_vs_fp32"] = len(set(a) & set(b))
    # round 5: WHERE the score error comes from - the cosine clip score alone (in logits: / tau), the rest of the score (the
    # scoring head's mean-cosine branch), and the feature error projected on the text direction (d . t^ / |f|: the first-order
    # term of the cosine's error) - so a HIP-vs-reference difference can be attributed
    r["clip_eps_logits"] = float((c16 - c32).abs().max() / tau)
    r["head_eps_logits"] = float(((s16 - c16 / tau) - (s32 - c32 / tau)).abs().max())
    th = torch.nn.functional.normalize(txt.float(), dim=-1)[0]
    r["proj_err_max"] = float((((f16 - f32) @ th) / f32.norm(dim=-1)).abs().max())
    return r


def gen_noise(_out):
    """How far the REFERENCE'S OWN precision sits from fp32 (the reference runs CLIP and the scoring head in bf16:
    mp_tools/vlmeval/vlm/gen_id_tspo.py:55, src/open_tspo/trainer/tspo_trainer.py:201): the installed transformers CLIP and the
    imported MultiModal_Align, cast to torch.bfloat16 exactly as `from_pretrained(torch_dtype=torch.bfloat16)` does, run on the
    CPU on the SAME weights / pixels / text as the HIP-vs-oracle tests (tests/golden/inputs.py: ENCODE_SCENARIOS, E2E_SCENARIOS).
    The fp32 side carries the bf16-ROUNDED parameters (what both the bf16 model and the HIP encoder hold), so the deltas are
    arithmetic noise only.  Stored: feature error as a fraction of the fp32 feature range, the smallest per-frame cosine, the
    score error in logits (round 5: split into its clip-score and scoring-head parts), and how many of the reference's own bf16
    top-k indices differ from its fp32 ones.  The HIP tolerances are tied to these (tests assert HIP-vs-fp32 <= 1.5 x per case,
    median over the three heavy-tailed videos <= 1.25 x) -> tests/golden/bf16_noise.json.  ~20 min of CPU."""
    import json
    from inputs import (ENCODE_SCENARIOS, E2E_SCENARIOS, E2E_TAU, E2E_WINDOW, clip_l14_state, e2e_video, e2e_selector_state,
                        e2e_texts, e2e_video_seed)
    g = {"clip": {}, "selector": {}, "encode": {}, "e2e": {}}
    only_e2e = os.environ.get("TSPO_NOISE_ONLY_E2E") == "1"     # regenerate the end-to-end group only, keep the rest of the file
    only_new = os.environ.get("TSPO_NOISE_ONLY_NEW") == "1"      # add the end-to-end scenarios the file does not hold yet, keep everything else
    only_e2e = only_e2e or only_new
    if only_e2e:
        g = json.load(open(os.path.join(HERE, "bf16_noise.json")))
        if not only_new:
            g["e2e"] = {}

    # small models: the golden CLIP cases
    for tag, cfgd, n in ([] if only_e2e else CLIP_CASES[:2]):
        for wname, state in (("normal", synth.clip_vision_state(**cfgd)), ("heavy_tailed", synth.clip_vision_state_heavy_tailed(cfgd))):
            f32, f16 = _both(_clip_model(cfgd, state), torch.from_numpy(clip_pixels(cfgd, n)[1]))
            g["clip"][f"{tag}.{wname}"] = dict(_feat_stats(f32, f16), frames=n)
            print(tag, wname, g["clip"][f"{tag}.{wname}"], flush=True)
    # CLIP-L/14 on the pixels of the HIP encode tests
    for name, (wname, n, seed) in ({} if only_e2e else ENCODE_SCENARIOS).items():
        f32, f16 = _both(_clip_model(synth.CLIP_L14, clip_l14_state(wname)), _normalize_u8(synth.uniform_u8((n, 3, 224, 224), seed)))
        g["encode"][name] = dict(_feat_stats(f32, f16), frames=n)
        print("encode", name, g["encode"][name], flush=True)
    # the whole pipeline (bf16 encode -> bf16 scoring head) on the end-to-end test's videos, planted-scene and independent text
    sel = e2e_selector_state()
    for name, (wname, n, needles) in E2E_SCENARIOS.items():
        if name in g["e2e"]:
            continue
        f32, f16 = _both(_clip_model(synth.CLIP_L14, clip_l14_state(wname)), _normalize_u8(e2e_video(n, needles, e2e_video_seed(name))))
        per_text = {tn: _head_noise(f32, f16, tx, sel, E2E_WINDOW, E2E_TAU) for tn, tx in e2e_texts(f32, needles).items()}
        g["e2e"][name] = {"frames": n, "tau": E2E_TAU, "features": _feat_stats(f32, f16), "texts": per_text,
                          "max_score_eps_logits": max(v["score_eps_logits"] for v in per_text.values())}
        print("e2e", name, g["e2e"][name], flush=True)
    for name, T, D, H, w, tau, M, ks in ([] if only_e2e else SELECTOR_CASES):
        img, txt, clip, state = selector_inputs(name, T, D, M)
        with torch.no_grad():
            s32, _ = ref_selector(D, H, state)(torch.from_numpy(img), torch.from_numpy(txt), torch.from_numpy(clip), window_size=w, score_tau=tau)
            m16 = ref_selector(D, H, state).to(torch.bfloat16)
            s16, _ = m16(torch.from_numpy(img).to(torch.bfloat16), torch.from_numpy(txt).to(torch.bfloat16),
                         torch.from_numpy(clip).to(torch.bfloat16), window_size=w, score_tau=tau)
        eps = float((s16.float() - s32).abs().max())
        g["selector"][name] = {"T": T, "D": D, "tau": tau, "score_eps_logits": eps, "score_eps_over_absmax": eps / float(s32.abs().max())}
    path = os.path.join(HERE, "bf16_noise.json")
    with open(path, "w") as f:
        json.dump(g, f, indent=1)
    print(f"noise: -> {path}")


def gen_full1024(out):
    """BASELINE.json configs[1] at FULL size through the reference's own chain (model/temporal_agent.py:177-192: extract_feature ->
    temporal_sampling -> inference_ts; mp_tools/vlmeval/vlm/gen_id_tspo.py:55 for the bf16 production dtype): 1024 frames of the
    full-size block video (tests/golden/inputs.py: full_video) -> installed transformers CLIP-L/14 (fp32 arithmetic on the bf16-rounded parameters) -> cosine clip
    score -> imported MultiModal_Align -> TSPOModel.inference_ts top-32 / top-64 / bin-max-32, for the planted-scene text (stored:
    it is built from the fp32 features) and an independent N(0,1) text; and the SAME chain in bf16, so the fixture also holds the
    reference's own score noise at this size.  A few KB of arrays -> tests/golden/full1024.npz.  ~25 min of CPU (8 threads)."""
    from inputs import (FULL_T, FULL_NEEDLES, FULL_K, E2E_TAU, E2E_WINDOW, clip_l14_state, full_video, e2e_selector_state,
                        e2e_independent_text)
    tspo_like = TSPOModel.inference_ts
    u8 = full_video()
    cache = os.environ.get("TSPO_FULL1024_CACHE")       # generation convenience: the two feature matrices (25 min of CPU) kept in a scratch file
    if cache and os.path.exists(cache):
        zz = np.load(cache)
        f32, f16 = torch.from_numpy(zz["f32"]), torch.from_numpy(zz["f16"])
    else:
        f32, f16 = _both(_clip_model(synth.CLIP_L14, clip_l14_state("normal")), _normalize_u8(u8), progress="full1024")
        if cache:
            np.savez(cache, f32=f32.numpy(), f16=f16.numpy())
    sel = e2e_selector_state()
    # the "question" about the planted scene: what distinguishes its frames from the average frame (as in e2e_texts)
    planted = torch.nn.functional.normalize(f32[FULL_NEEDLES].mean(0, keepdim=True) - f32.mean(0, keepdim=True), dim=-1)
    texts = {"planted": planted, "independent": torch.from_numpy(e2e_independent_text(0))}
    st = _feat_stats(f32, f16)
    out["feat.err_over_range_bf16"] = np.array(st["err_over_range"])
    out["feat.min_cos_bf16"] = np.array(st["min_cos"])
    out["feat.range"] = np.array(st["range"])
    rows = [0, 1, 100, 511, 900, 1023]
    out["feat.rows"] = np.array(rows, np.int64)
    out["feat.values"] = f32[rows].numpy()
    out["feat.row_norms"] = f32.norm(dim=-1).numpy()
    out["feat.sum"] = np.array([f32.double().sum().item(), f32.double().abs().sum().item()])
    out["u8.checksum"] = np.array([int(u8.astype(np.uint64).sum()), int(u8[::97].astype(np.uint64).sum())], np.uint64)
    for tn, tx in texts.items():
        s32, s16, c32, c16 = _head_both(f32, f16, tx, sel, E2E_WINDOW, E2E_TAU)
        out[f"{tn}.text"] = tx.numpy()
        out[f"{tn}.scores"] = s32.numpy()
        out[f"{tn}.clip"] = c32.numpy()
        out[f"{tn}.scores_bf16"] = s16.numpy()
        for k in (len(FULL_NEEDLES), FULL_K, 64):
            out[f"{tn}.topk{k}"] = quiet(tspo_like, None, s32, "topk", k)[0].numpy()
            out[f"{tn}.topk{k}_bf16"] = quiet(tspo_like, None, s16, "topk", k)[0].numpy()
        out[f"{tn}.binmax{FULL_K}"] = quiet(tspo_like, None, s32, "bin-max", FULL_K)[0].numpy()
        out[f"{tn}.binmax{FULL_K}_bf16"] = quiet(tspo_like, None, s16, "bin-max", FULL_K)[0].numpy()
        print(f"full1024 {tn}: eps(bf16 vs fp32) = {float((s16 - s32).abs().max()):.4f} logits, spread {float(s32.max() - s32.min()):.2f}; "
              f"top-32 overlap {len(set(out[f'{tn}.topk32'].tolist()) & set(out[f'{tn}.topk32_bf16'].tolist()))}/32", flush=True)


def main():
    groups = {"selector": gen_selector, "misc": gen_misc, "gumbel": gen_gumbel, "train": gen_train, "train_full": gen_train_full, "clip": gen_clip,
              "glue": gen_glue, "published": gen_published, "noise": gen_noise, "full1024": gen_full1024}
    which = sys.argv[1:] or list(groups)
    for g in which:
        out = {}
        groups[g](out)
        if g in ("glue", "published", "noise"):
            continue          # write their own JSON files (strings, nested lists)
        path = os.path.join(HERE, f"{g}.npz")
        np.savez_compressed(path, **out)
        print(f"{g}: {len(out)} arrays -> {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
