"""`python bench.py --gpus N` must start N ranks by itself (the driver's N=1 call and its torch.distributed.run call both
keep working).  The rank body is stubbed (`--stub-body`): gloo on the CPU, no GPU, so this runs in the build container."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=300)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_self_launches_two_ranks_and_relays_one_json_line():
    res = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-body", "ok")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3 and lines[0]["warmup"] == 1
    assert lines[0]["value"] == 3.0  # 1 + 2: both ranks took part in the all-reduce


def test_failing_rank_gives_nonzero_exit_code():
    res = _run("--gpus", "2", "--stub-body", "fail")
    assert res.returncode != 0
    assert not _json_lines(res.stdout)


def test_world_8_rehearsal_at_the_drivers_shape():
    """What an 8-GPU node meets first, on gloo/CPU: `bench.py --gpus 8` under its own launcher — eight ranks, the real
    SlabExchange (one whole-slab all-reduce per step), the replica-signature check, ONE JSON line whose `config` reports the
    process group's own world size and backend."""
    res = _run("--gpus", "8", "--steps", "3", "--warmup", "1", "--stub-body", "ok")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    line = lines[0]
    assert line["n_gpus"] == 8 and line["value"] == 36.0 and line["scaling"] == "weak"  # 1 + … + 8: all ranks in the all-reduce
    assert line["config"] == {"parallelism": "dp8", "rccl_ranks": 8, "backend": "gloo"}


def test_world_8_diverged_replica_ends_the_run_with_exit_code_4():
    """A replica that trained on something else (here: the last rank's step 1 never went through the exchange) must end the
    run with the divergence code instead of a number."""
    res = _run("--gpus", "8", "--steps", "3", "--warmup", "1", "--stub-body", "diverge")
    assert res.returncode != 0
    assert not _json_lines(res.stdout)
    assert "LoRA slab differs across ranks" in res.stderr and "exitcode  : 4" in res.stderr  # (torchrun's failure report)


def test_eight_ranks_refuse_to_share_the_visible_devices():
    """One process per GPU: on a node that shows fewer devices than local ranks (this container shows none) every rank of the
    REAL body exits with code 5 before anything touches a device — two ranks time-slicing one GPU must never be reported as
    N GPUs; `--shared-gpu` is the explicit rehearsal switch."""
    import torch

    if torch.cuda.device_count() >= 8:
        import pytest

        pytest.skip("a full node: nothing to refuse")
    res = _run("--gpus", "8", "--steps", "1", "--warmup", "0", "--no-extra", "--no-cpu-baseline")
    assert res.returncode != 0
    assert not _json_lines(res.stdout)
    assert "visible GPU(s) for 8 ranks" in res.stderr and "--shared-gpu" in res.stderr


def test_single_rank_runs_in_process():
    res = _run("--gpus", "1", "--stub-body", "ok")
    assert res.returncode == 0, res.stderr[-2000:]
    assert _json_lines(res.stdout)[0]["n_gpus"] == 1
    assert "launcher:" not in res.stderr


def test_bench_workloads_follow_baseline_configs():
    """bench.CONFIGS mirrors BASELINE.json:configs (index = position in that list; 1 is the CPU case) and the synthetic batches
    have the shapes each config names: prior preservation doubles the rows (instance then class), config 3 feeds token ids,
    config 5 is the SD2.1-768 shape; noise and timesteps are rank-invariant, latents / conditioning are sharded."""
    import json

    import torch

    import bench

    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert sorted(bench.CONFIGS) == list(range(2, len(base["configs"]) + 1))
    assert "rank=4" in base["configs"][1] and bench.CONFIGS[2]["rank"] == 4 and bench.CONFIGS[2]["batch"] == 4
    assert "rank=8" in base["configs"][2] and bench.CONFIGS[3]["rank"] == 8 and bench.CONFIGS[3]["text_encoder"]
    assert "prior-preservation" in base["configs"][3] and bench.CONFIGS[4]["prior"]
    assert "rank=16" in base["configs"][4] and bench.CONFIGS[5]["rank"] == 16 and bench.CONFIGS[5]["latent"] == 96
    c4 = bench.CONFIGS[4]
    d0 = bench.synthetic_steps(2, c4["batch"], c4["latent"], 0, 2, "cpu", rows_per_image=2)
    d1 = bench.synthetic_steps(2, c4["batch"], c4["latent"], 1, 2, "cpu", rows_per_image=2)
    lat, noise, t, ctx = d0[0]
    assert lat.shape == (8, 4, 64, 64) and noise.shape == lat.shape and t.shape == (8,) and ctx.shape == (8, 77, 768)
    assert torch.equal(d0[0][1], d1[0][1]) and torch.equal(d0[0][2], d1[0][2])  # noise / timesteps: the same on every rank
    assert not torch.equal(d0[0][0], d1[0][0]) and not torch.equal(d0[0][3], d1[0][3])  # latents / text: each rank its shard
    c5 = bench.CONFIGS[5]
    lat5, _, _, ctx5 = bench.synthetic_steps(1, c5["batch"], c5["latent"], 0, 1, "cpu", c5["ctx_len"], c5["ctx_dim"])[0]
    assert lat5.shape == (1, 4, 96, 96) and ctx5.shape == (1, 77, 1024)
    ids = bench.synthetic_steps(1, 4, 64, 0, 1, "cpu", ids=True)[0][3]
    assert ids.shape == (4, 77) and ids.dtype == torch.int64 and int(ids[:, 0].min()) == 49406 and int(ids[:, -1].max()) == 49407
    # config 2's tensors are what they were before the other configs existed (the headline's inputs did not move)
    g = torch.Generator().manual_seed(1000)
    want = torch.randn(4, 4, 64, 64, generator=g) * 0.18215
    assert torch.equal(bench.synthetic_steps(1, 4, 64, 0, 1, "cpu")[0][0], want)


# the distinct LoRA linears of config 2 (SD1.5, batch 4, 64² latents, 77-token context) as (M, K, N, count)
SURVEY_CFG2_LAYERS = [(16384, 320, 320, 30), (16384, 320, 2560, 5), (308, 768, 320, 10), (4096, 640, 640, 30), (4096, 640, 5120, 5),
                      (308, 768, 640, 10), (1024, 1280, 1280, 30), (1024, 1280, 10240, 5), (308, 768, 1280, 12), (256, 1280, 1280, 6),
                      (256, 1280, 10240, 1)]


def test_contract_bytes_of_config_2_are_surveys_5331_mb():
    """`roofline.step_level.algorithmic_MB_per_step_survey` is SURVEY §8(d)'s figure: 5 331 MB per step at config 2 (forward
    2 359 + backward 2 972, a dX for every layer but the frozen encoder's attn2.to_k / to_v), whatever the step itself skips —
    it skips the dX of the first block's q/k/v (their input has no trainable producer): 32 MB less, reported beside it."""
    from diffusion_finetuning_amd.trainer import survey_bytes_flops

    rows, skipped = [], 0
    for M, K, N, count in SURVEY_CFG2_LAYERS:
        ctx = K == 768
        for _ in range(count):
            first_qkv = (M, K, N) == (16384, 320, 320) and skipped < 3  # down_blocks.0 ... attn1.to_q/k/v
            skipped += first_qkv
            rows.append((M, K, N, 4, False, not ctx and not first_qkv, ctx))
    assert len(rows) == 144
    c = survey_bytes_flops(rows, 2, contract=True)
    assert abs(c["fwd_bytes"] / 1e6 - 2358.8) < 0.1 and abs(c["bwd_bytes"] / 1e6 - 2972.3) < 0.1
    assert abs((c["fwd_bytes"] + c["bwd_bytes"]) / 1e6 - 5331.1) < 0.1
    ran = survey_bytes_flops(rows, 2)
    assert abs((ran["fwd_bytes"] + ran["bwd_bytes"]) / 1e6 - 5299.0) < 0.3  # what rounds 4–5 reported as the contract
    assert abs((c["fwd_flops"] + c["bwd_flops"]) / 1e9 - 1476) < 2  # SURVEY: 1 476 GF per step
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "survey_work(2 if args.dtype != \"f32\" else 4, contract=True)" in src  # the line's figure is the contract's


def test_launch_floor_fields_of_the_roofline_object():
    """`roofline.step_level` carries the measured launch floor (empty kernels at every hot-path launch site, in the step's order)
    and the HBM fraction against the kernel time above that floor."""
    import bench

    gemm = "lora_gemm_kernel<*, 64, 64|128|160, true>"
    roof = {"step_level": {"hbm_bound_ms": 0.666, "kernel_ms_per_step": 4.0},
            "fused_gemm_classes": {gemm: {"launches_per_step": 110.0, "ms_per_step": 1.9}}}
    floor = {gemm: {"launches": 110.0, "ms": 0.55}, "lora_grad_{mfma_,}kernel, ranks <= 4": {"launches": 1.0, "ms": 0.02},
             "attn_flash_fwd_kernel": {"launches": 16.0, "ms": 0.1}}  # (attention kinds are not §8(d) layers)
    bench.add_launch_floor(roof, floor, tot_ms=4.0)
    sl = roof["step_level"]
    assert abs(sl["launch_floor_ms"] - 0.57) < 1e-9 and sl["launch_floor_launches_per_step"] == 111.0
    assert abs(sl["frac_vs_launch_bounded"] - 0.666 / (4.0 - 0.57)) < 1e-9
    assert abs(sl["launch_floor_all_library_kinds_ms"] - 0.67) < 1e-9
    assert abs(sl["frac_ceiling_with_per_layer_launches"] - 0.666 / (0.666 + 0.57)) < 1e-9
    assert abs(roof["fused_gemm_classes"][gemm]["launch_floor_us_per_launch"] - 5.0) < 1e-9
    bench.add_launch_floor(roof2 := {"step_level": {}}, {}, 4.0)  # no floor pass: nothing added, nothing raised
    assert roof2 == {"step_level": {}}
