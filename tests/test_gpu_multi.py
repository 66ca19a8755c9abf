"""GPU tests of the multi-rank leg as it is LAUNCHED (BASELINE configs[3] / configs[4]; reference: train_deepspeed.sh:14-16
`torchrun --nproc_per_node`, scripts/zero3.json:14-33, tspo_trainer.py:610-634):

* `python bench.py --gpus 2 --same-device --backend gloo` from a bare shell (no WORLD_SIZE): bench.py spawns its own
  ranks, prints ONE JSON line with n_gpus == 2 and a `comm` block saying how many ranks the collective library saw;
* backend "nccl" (= RCCL) really executes on the leased GPU: a one-rank group, `all_reduce` of the trainer's gradient
  bucket view and `all_gather_into_tensor` of a [64,768] feature block (librccl loads, device binding, bucket alignment);
* frame-sharded encode (`FrameScorer.encode(shard_frames=True)`) with the real CLIP-L/14 encoder over two ranks ==
  the single-process encode, bit for bit;
* `python -m tspo_amd.train --gpus 2` spawns its own ranks; the feature-cache training flow of configs[0]
  (toy_example.sh: cached CLIP features -> G rollouts -> update) through the CLI, including resume.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tspo_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _ok(r, tag):
    """returncode == 0, else the child's whole stderr goes to gpurun_out/failed_<tag>.stderr (merged back by gpurun: a failure that
    shows once in many runs can be read afterwards) and its tail into the assertion message."""
    if r.returncode != 0:
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"failed_{tag}.stderr"), "w") as f:
                f.write(r.stderr + "\n---- stdout ----\n" + r.stdout)
        except OSError:
            pass
    assert r.returncode == 0, r.stderr[-3000:]


def _json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def test_bench_self_spawns_two_ranks_from_a_bare_shell():
    """The command form the driver uses for N=1, with N=2 and no launcher environment."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo",
                        "--frames", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pruned"],
                       env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    _ok(r, "bench_self_spawn_2ranks")
    lines = [l for l in _json_lines(r.stdout) if "metric" in l]
    assert len(lines) == 1, r.stdout[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["launcher"] == "self-spawn" and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and line["rollouts_per_s"] > 0 and line["steps"] == 2
    lf = line["configs4_long_form_sharded"]            # N > 1: the strong-scaling point of configs[4] rides in the default launch
    assert lf and "error" not in lf and lf["frames_per_video"] == 4096 and lf["frames_per_gpu"] == 2048 and lf["frames_scored_per_s"] > 0
    assert lf["scaling"] == "strong" and lf["workload"].startswith("configs[4] long-form option")
    by_rank = line["ms_per_step_by_rank"]             # every rank's own clock over the timed region: a slow rank must be visible
    assert len(by_rank) == 2 and all(t > 0 for t in by_rank) and abs(max(by_rank) - line["ms_per_step"]) < 1e-3 * line["ms_per_step"] + 1e-3
    comm = line["comm"]
    assert comm["backend"] == "gloo" and comm["world"] == 2 and comm["allreduce_us"] > 0 and comm["sum_correct"] is True
    assert "error" not in comm


def test_bench_shard_frames_two_ranks_strong_scaling_line():
    """configs[4]'s long-form option as bench.py launches it (`--shard-frames`): ONE video per step, its frames sharded over the
    ranks for the encode, one all-gather, replicated scoring head + top-k.  Two ranks on the one GPU (gloo, host-staged
    all-gather): a STRONG-scaling line (value counts the ONE video's frames, not frames x ranks), the workload named from
    the arguments, every rank encoding half of the frames (that the sharded features equal the single-process ones bit for
    bit is test_frame_sharded_encode_real_encoder_two_ranks below); and the unsharded line of the same size for comparison."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
            "--no-pruned", "--no-720p", "--no-rollouts", "--no-comm-probe"]
    r = subprocess.run(base + ["--gpus", "2", "--same-device", "--backend", "gloo", "--shard-frames"],
                       env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    _ok(r, "bench_shard_frames_2ranks")
    line = [l for l in _json_lines(r.stdout) if "metric" in l][0]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["launcher"] == "self-spawn"
    cfg = line["config"]
    assert cfg["workload"].startswith("configs[4] long-form option") and "T=128" in cfg["workload"] and "2 rank(s)" in cfg["workload"]
    assert cfg["frames_per_gpu_per_step"] == 64 and cfg["videos_per_step_total"] == 1 and "sharded" in cfg["parallelism"]
    # value = the ONE video's frames / s: 128 frames x 2 steps over the timed region
    assert abs(line["value"] - 128 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-2 * line["value"]
    assert line["roofline"]["frames_per_launch"] == 64           # rank 0's GEMM launches process its own half
    one = subprocess.run(base + ["--gpus", "1"], env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    _ok(one, "bench_unsharded_128")
    l1 = [l for l in _json_lines(one.stdout) if "metric" in l][0]
    assert l1["scaling"] == "weak" and l1["config"]["workload"].startswith("custom:") and "T=128" in l1["config"]["workload"]
    assert l1["roofline"]["frames_per_launch"] == 128


def test_bench_under_torch_distributed_run():
    """The driver's launch form for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`), here with two ranks on the one GPU (gloo), and the N = 1 form under the launcher."""
    from tspo_amd.dist import free_port
    for n, extra in ((2, ["--same-device", "--backend", "gloo"]), (1, [])):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
                            "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n),
                            "--frames", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pruned", "--no-720p"] + extra,
                           env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
        _ok(r, "bench_torchrun")
        lines = [l for l in _json_lines(r.stdout) if "metric" in l]
        assert len(lines) == 1, r.stdout[-2000:]
        line = lines[0]
        assert line["n_gpus"] == n and line["value"] > 0 and line["comm"]["world"] == n and "error" not in line["comm"]
        assert line["launcher"] == ("torch.distributed.run" if n > 1 else "single process")


def _nccl_worker(q):
    import datetime
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from tspo_amd import dist as td, ops as o
    from tspo_amd.pipeline import PolicyTrainer
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        os.environ["MASTER_PORT"] = str(td.free_port())
        rank, world, _ = td.init_from_env("nccl", device=dev, force=True)
        assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
        D = 768
        flat = torch.randn(o.flat_offsets(D)["__total__"][0], device=dev)
        tr = PolicyTrainer(flat, dim=D)
        g = torch.Generator(device=dev).manual_seed(3)
        tr.grad.copy_(torch.randn(tr.grad.shape, generator=g, device=dev))
        before = tr.grad.clone()
        view = tr.grad[: tr.n_train]
        assert view.data_ptr() % 16 == 0 and view.is_contiguous()
        dist.all_reduce(view, op=dist.ReduceOp.SUM)                                # the call dist.py issues over RCCL
        assert td.allreduce_bucket_(tr.grad, tr.n_train) == 1                      # ... and through the product's wrapper
        torch.cuda.synchronize()
        assert torch.equal(tr.grad, before)                                        # sum over one rank; ffn_o tail untouched
        # a view that does NOT start at the allocation base (16-byte offset) goes through RCCL as well
        off = tr.grad[4: 4 + 4096]
        dist.all_reduce(off, op=dist.ReduceOp.SUM)
        # optimizer_step() = reduce_fn (RCCL) + clip + AdamW
        tr._micro = 1
        st = tr.optimizer_step()
        assert st["world"] == 1 and torch.isfinite(tr.flat).all()
        # frame-sharded encode's exchange: all_gather_into_tensor of a [64,768] feature block
        feats = torch.randn(64, 768, generator=g, device=dev)
        out = torch.empty(64, 768, device=dev)
        dist.all_gather_into_tensor(out, feats)
        assert torch.equal(out, feats)
        assert torch.equal(td.sharded_apply(lambda t: t * 2, feats), feats * 2)
        # packed metrics all-reduce (tspo_trainer.py:610-634 -> one collective)
        m = td.reduce_metrics(td.pack_metrics({"reward": 0.5, "loss": 2.0}, [1.0, 0.25]), 2, ["a", "b"])
        assert abs(m["reward"] - 0.5) < 1e-12 and abs(m["rewards/b"] - 0.25) < 1e-12
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
        t1.record()
        torch.cuda.synchronize()
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", ver, t0.elapsed_time(t1) / 10 * 1e3))
    except Exception as e:      # noqa: BLE001 - report to the parent
        import traceback
        q.put(("fail", traceback.format_exc(), 0.0))


def test_rccl_backend_executes_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(q,))
    p.start()
    status, info, us = q.get(timeout=600)
    p.join(timeout=120)
    assert status == "ok", info
    assert p.exitcode == 0
    print(f"\n[rccl {info}] one-rank all-reduce of the 11.8 MB gradient bucket: {us:.1f} us")


def _shard_worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    from tspo_amd import dist as td, ops as o
    from tspo_amd.pipeline import FrameScorer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    td.init_from_env("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    c = bench.CLIP_L14
    clipw = o.ClipVitWeights(bench.random_clip_state(c, dev), c, dev)
    scorer = FrameScorer(clipw, bench.flat_from_state(bench.random_selector_state(768, dev), 768, dev))
    px = torch.randint(0, 256, (1, n_frames, 3, 224, 224), generator=torch.Generator(device=dev).manual_seed(5), device=dev,
                       dtype=torch.uint8)
    feats = scorer.encode(px, shard_frames=True)                 # each rank encodes n/2 frames; one all-gather
    q.put((rank, feats.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_encode_real_encoder_two_ranks():
    """configs[4] / SURVEY C3: the frames of ONE video split over the ranks, CLIP-L/14 on each share, features
    all-gathered -> bitwise the single-process encode of all frames (frames are independent; same kernels per frame)."""
    import bench
    from tspo_amd.pipeline import FrameScorer
    n, world = 128, 2
    from tspo_amd.dist import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    c = bench.CLIP_L14
    clipw = ops.ClipVitWeights(bench.random_clip_state(c, DEV), c, DEV)
    scorer = FrameScorer(clipw, bench.flat_from_state(bench.random_selector_state(768, DEV), 768, DEV))
    px = torch.randint(0, 256, (1, n, 3, 224, 224), generator=torch.Generator(device=DEV).manual_seed(5), device=DEV,
                       dtype=torch.uint8)
    want = scorer.encode(px).cpu().numpy()
    assert want.shape == (1, n, 768) and np.isfinite(want).all()
    for rank, got in res:
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want, err_msg=f"rank {rank}")


def _write_feature_caches(root, n_files, T, D=768):
    """Feature caches in the evaluation flow's format (tspo_amd.io.save_feature_cache: what FrameIdGenerator writes on a
    miss, gen_id_tspo.py:80-86) + the training-side extras FeatureCacheDataset understands ("mask", "type")."""
    from tspo_amd import io as tio
    g = torch.Generator().manual_seed(77)
    for i in range(n_files):
        t = T + 16 * i                                        # videos differ in length
        txt = torch.randn(1, D, generator=g)
        img = torch.randn(t, D, generator=g)
        mask = torch.zeros(t, dtype=torch.bool)
        mask[10 + 5 * i: 10 + 5 * i + t // 6] = True
        img = img + 0.1 * mask[:, None] * txt
        clip = torch.nn.functional.cosine_similarity(img, txt, dim=-1)
        path = tio.feature_cache_path(root, "toy", i)
        tio.save_feature_cache(path, img.to(torch.bfloat16), txt.to(torch.bfloat16), clip.to(torch.bfloat16),
                               torch.arange(0, 30 * t, 30))
        stat = torch.load(path)
        stat["mask"], stat["type"] = mask, ("general" if i == 2 else "specific")
        torch.save(stat, path)


def _run_train_cli(args, timeout=900):
    r = subprocess.run([sys.executable, "-m", "tspo_amd.train"] + args, env=_clean_env(), cwd=ROOT, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return _json_lines(r.stdout)


def test_train_cli_from_feature_cache_with_resume(tmp_path):
    """configs[0] (toy_example.sh): pre-extracted CLIP features -> `python -m tspo_amd.train --features` -> G=4 rollouts
    per prompt, accumulation 2, checkpoints; an interrupted + resumed run ends with the same parameters as an
    uninterrupted one (the data stream resumes at the right file without re-reading the skipped ones)."""
    from tspo_amd import io as tio
    root = str(tmp_path / "feats")
    _write_feature_caches(root, 3, 96)
    common = ["--features", root, "--num-generations", "4", "--training-sample-len", "8", "--gradient-accumulation-steps", "2",
              "--save-steps", "1", "--max-steps", "3", "--seed", "5"]
    a = str(tmp_path / "a")
    la = _run_train_cli(common + ["--output-dir", a])
    assert [l["step"] for l in la] == [1, 2, 3]
    assert all(np.isfinite(l["loss"]) and np.isfinite(l["grad_norm"]) for l in la) and max(l["grad_norm"] for l in la) > 0
    assert max(l["reward_std"] for l in la) > 0
    assert la[0]["ts_length"] == 8 and la[1]["ts_length"] == 6     # step 2 = files (2: general -> k/2, 0: specific): (4 + 8) / 2
    b = str(tmp_path / "b")
    from tspo_amd import train as tt
    # first leg stops after step 2 (own process), second leg resumes from checkpoint-2
    r = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r)\n"
                        "from tspo_amd import train as tt\n"
                        "cfg = tt.TrainConfig(output_dir=%r, max_steps=3, num_generations=4, training_sample_len=8, "
                        "gradient_accumulation_steps=2, save_steps=1, seed=5)\n"
                        "tt.train(cfg, tt.FeatureCacheDataset(%r), resume=False, stop_after=2)\n" % (ROOT, b, root)],
                       env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    _ok(r, "train_cli_2ranks")
    lb = _run_train_cli(common + ["--output-dir", b])
    assert [l["step"] for l in lb] == [3]
    sa = tio.load_selector_safetensors(os.path.join(a, "checkpoint-3", "model.safetensors"))
    sb = tio.load_selector_safetensors(os.path.join(b, "checkpoint-3", "model.safetensors"))
    assert sorted(sa) == sorted(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    s0 = tio.load_selector_safetensors(os.path.join(a, "checkpoint-1", "model.safetensors"))
    assert any(not torch.equal(sa[k], s0[k]) for k in sa)         # the run moved the parameters
    assert tt.FeatureCacheDataset(root).files and len(tt.FeatureCacheDataset(root).files) == 3


def test_train_cli_self_spawns_two_ranks(tmp_path):
    """`python -m tspo_amd.train --gpus 2` with no launcher: two ranks (gloo, both on the one GPU), one metrics file,
    rank 0's checkpoints, and a clean exit of BOTH ranks after the final save (barrier + destroy in train())."""
    out = str(tmp_path / "run")
    lines = _run_train_cli(["--gpus", "2", "--backend", "gloo", "--dim", "64", "--frames", "96", "--num-generations", "4",
                            "--training-sample-len", "8", "--max-steps", "2", "--save-steps", "1", "--output-dir", out,
                            "--per-device-train-batch-size", "2"])
    assert [l["step"] for l in lines] == [1, 2]                               # rank 0 alone prints / logs
    logged = [json.loads(l) for l in open(os.path.join(out, "metrics.jsonl"))]
    assert [l["step"] for l in logged] == [1, 2]
    assert sorted(d for d in os.listdir(out) if d.startswith("checkpoint-")) == ["checkpoint-1", "checkpoint-2"]
    assert os.path.exists(os.path.join(out, "checkpoint-2", "model.safetensors"))


def test_bench_refuses_more_ranks_than_gpus_without_hanging():
    """`python bench.py --gpus 2` on a box with ONE visible GPU (no --same-device): a clear message and a non-zero exit code
    at once - not a rendezvous that waits for a rank that can never get a device (what the first real N > 1 run must not do
    when a node comes up with fewer GPUs than asked for)."""
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one visible GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 but only 1 GPU(s) visible" in r.stderr and not _json_lines(r.stdout)


def test_self_spawn_quotes_the_failing_ranks_stderr(tmp_path):
    """A rank that dies takes the others down and the launcher's own message carries the tail of ITS stderr."""
    from tspo_amd import dist as tdist
    prog = tmp_path / "prog.py"
    prog.write_text("import os, sys, time\n"
                    "if os.environ['RANK'] == '1':\n"
                    "    sys.stderr.write('rank one says: no device\\n'); sys.exit(7)\n"
                    "time.sleep(60)\n")
    code = ("import sys; sys.path.insert(0, %r); from tspo_amd import dist as d; sys.exit(d.self_spawn(2, [%r], timeout=50))"
            % (ROOT, str(prog)))
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 7
    assert "rank 1 of 2 exited with code 7" in r.stderr and "rank one says: no device" in r.stderr.split("Last lines of its stderr")[-1]


def test_bench_line_explains_itself_and_times_the_dp_path():
    """The default single-GPU bench line (short run): per-step spread, the clock / power sampled during the timed steps with the
    roofline fraction at that clock, and `rollouts_dp_path` - the reference's own training configuration (one prompt per
    micro-step, two micro-steps, the bucket all-reduce really issued on a live one-rank RCCL group) with its launch count."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-pruned", "--no-720p"], env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    _ok(r, "bench_default_line")
    line = [l for l in _json_lines(r.stdout) if "metric" in l][0]
    assert line["ms_per_step_min"] <= line["ms_per_step_median"] <= line["ms_per_step_max"]
    assert abs(line["ms_per_step_median"] - line["ms_per_step"]) < 0.2 * line["ms_per_step"]
    assert line["config"]["workload"].startswith("configs[1]:") and "configs[2]" in line["rollouts_config"]["workload"]
    roof = line["roofline"]
    assert roof["kernel"] == "gemm_bf16_a9_kernel" and 0.3 < roof["frac"] < 0.7
    # `traffic` is quoted only from a PMC summary bound to THIS library (else null + the reason) - never a stale round's bytes
    from tspo_amd.build import lib_identity
    me = lib_identity()
    if roof["traffic"] is None:
        assert "no profiles/" in roof["traffic_source"] and me["src_sha256"][:12] in roof["traffic_source"]
    else:
        tj = json.load(open(os.path.join(ROOT, roof["traffic_source"].split(" ")[0])))
        assert tj["library"]["lib_sha256"] == me["lib_sha256"] or tj["library"]["src_sha256"] == me["src_sha256"]
        assert roof["traffic_by_form"] and all(f["read_ratio"] > 0.9 for f in roof["traffic_by_form"])
    assert roof["gpu_state"]["samples"] > 0, roof["gpu_state"]
    assert 500 < roof["sclk_mhz"] <= 2400 and 100 < roof["power_w"] < 2000
    assert roof["frac_at_sustained_clock"] >= roof["frac"] - 1e-6
    assert "fused single-rank" in line["rollouts_variant"]
    dp = line["rollouts_dp_path"]
    assert "error" not in dp, dp
    assert dp["config"] == {"prompts_per_micro_step": 1, "grad_accum_steps": 2, "T": 512, "G": 8, "k": 16, "ranks": 1}
    assert dp["rollouts_per_s"] > 0 and "1-rank process group" in dp["allreduce"]
    by = dp["launches_by_kernel"]
    assert by is not None and dp["launches_per_optimizer_step"] == round(sum(by.values()))
    # (a one-rank RCCL all-reduce is a device copy, which the launch count leaves out like every memcpy; `comm` below shows the
    # group the step's reduce_fn ran on)
    assert by.get("adamw_clip_kernel") == 1.0 and by.get("gumbel_topk_kernel") == 1.0       # one update, ONE sampler launch for both micro-steps
    assert dp["launches_per_optimizer_step"] <= 16          # round 5: the window's two prompts as one stacked rollout / backward (sequential: 28)
    assert dp["sequential"]["launches_per_optimizer_step"] <= 28 and dp["sequential"]["rollouts_per_s"] < dp["rollouts_per_s"]
    assert isinstance(dp["runtime_copies_per_optimizer_step"], dict) and 0 < dp["roofline"]["frac"] < 1
    comm = line["comm"]
    assert comm["world"] == 1 and comm["backend"] == "nccl" and comm["sum_correct"] and comm["distinct_devices"] == 1
    assert comm["ranks"][0]["pci"] == roof["gpu_state"]["pci"]
