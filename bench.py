#!/usr/bin/env python3
"""Headline benchmark: images/s of one full SD1.5 LoRA train step (rank 4, batch 4/GPU, 512² → 64×64×4 latents,
fp16 storage/compute with fp32 accumulate and fp32 master LoRA), synthetic latents, 1..8 GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = add_noise → UNet forward (144 fused LoRA linears) → fused MSE → backward (fused dX + factor-grad
kernels) → [RCCL all-reduce of the 5 MB LoRA gradient slab] → fused clip + AdamW.  Inputs for every step are
resident in HBM before the timed region.  One JSON line is printed by rank 0 (contract: task statement ④).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}  # dense peaks, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per step (train_batch_size)")
    ap.add_argument("--rank-r", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64, help="latent height=width (512² images → 64)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--no-prof", action="store_true", help="do not attach kernel events in the timed region")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel from the host each step instead of replaying the recorded hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-noise", action="store_true",
                    help="feed pre-drawn noise / timesteps instead of drawing them on the device inside the step")
    ap.add_argument("--cpu-steps", type=int, default=6, help="timed oracle steps of the CPU baseline (~3 s each on 16 cores)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --shared-gpu rehearses N ranks on one GPU")
    ap.add_argument("--shared-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--stub-body", default=None, choices=["ok", "fail"],
                    help="test hook: ranks only rendezvous over gloo on the CPU and rank 0 prints a stub JSON line "
                         "('fail': rank 1 exits non-zero) — exercises the launcher without a GPU")
    return ap.parse_args()


def _free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` called directly (no WORLD_SIZE in the environment): start N fresh rank processes, one
    per GPU, the way the reference starts its trainer (`accelerate launch`, training_scripts/run_lora_db_unet_only.sh:6 —
    here `python -m torch.distributed.run`), relay rank 0's JSON line and propagate a non-zero exit code.  The parent
    never touches the GPU and never replaces itself: the ranks are children."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    log(f"launcher: starting {n} ranks: {' '.join(cmd[1:])}")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:  # rank 0's result line (and anything else the ranks print) goes to our stdout as it comes
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc != 0:
        log(f"launcher: ranks exited with code {rc}")
    return rc


def stub_body(args, rank, world):
    """Launcher test body (CPU only): every rank joins a gloo group and contributes to one all-reduce; rank 0 prints a
    line with the contract's keys.  'fail' makes rank 1 exit non-zero so the launcher's error propagation is visible."""
    import torch.distributed as dist

    if args.stub_body == "fail" and rank == 1:
        sys.exit(3)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": float(t.item()), "unit": "ranksum", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "stub": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def build_model(device, dtype, rank_r):
    import diffusion_finetuning_amd as dfa
    from harness.unet import UNet2DConditionModel, sd15_config

    torch.manual_seed(0)  # identical random-init weights on every rank (no checkpoints offline)
    with torch.device(device):
        unet = UNet2DConditionModel(sd15_config())
    unet = unet.to(dtype)
    unet.requires_grad_(False)  # train_lora_dreambooth.py:595
    dfa.inject_trainable_lora(unet, r=rank_r)  # :596-598
    from diffusion_finetuning_amd.attention import set_use_hip_geglu, set_use_memory_efficient_attention_xformers

    set_use_memory_efficient_attention_xformers(unet, True)  # :623-624 (--use_xformers): here the HIP attention core
    set_use_hip_geglu(unet, True)  # the fused GEGLU gate after each `proj` LoRA linear
    g = torch.Generator(device="cpu").manual_seed(1)
    with torch.no_grad():  # warm-started `up` so no kernel sees the all-zero branch
        for up, _ in dfa.extract_lora_ups_down(unet):
            up.weight.copy_((torch.randn(up.weight.shape, generator=g) * 0.01).to(device))
    return unet


def synthetic_steps(n_steps, batch, latent, rank, world, device):
    """Per-step inputs, resident in HBM: noise/timesteps are rank-invariant (set_seed semantics,
    train_lora_dreambooth.py:509-510); each rank owns its shard of the latents / text embeddings."""
    out = []
    for s in range(n_steps):
        g = torch.Generator().manual_seed(1000 + s)
        lat = torch.randn(world * batch, 4, latent, latent, generator=g) * 0.18215
        ctx = torch.randn(world * batch, 77, 768, generator=g)
        noise = torch.randn(batch, 4, latent, latent, generator=g)
        t = torch.randint(0, 1000, (batch,), generator=g)
        sl = slice(rank * batch, (rank + 1) * batch)
        out.append((lat[sl].to(device), noise.to(device), t.to(device), ctx[sl].to(device)))
    return out


def measured_traffic(kernel_name, dtype):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE
    doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950).  PMC counters cannot be read from inside the
    process, so this is the newest committed measurement — and only if it was taken from the very sources the
    loaded library was built from (digest of csrc/ + header, stamped by tools/summarize_profile.py); otherwise null."""
    import glob
    import re

    from diffusion_finetuning_amd import build_native

    m = re.search(r"<\*, (\d+), ([\d|]+), (true|false)>", kernel_name)  # a profiler class may cover several tile widths
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files or dtype != "f16":
        return None
    table = json.load(open(files[-1]))
    stamp = os.path.join(build_native.LIB_DIR, "liblora_hip.stamp")
    built_from = open(stamp).read().strip() if os.path.exists(stamp) else None
    if not built_from or table.get("_csrc_digest") != built_from:
        return None
    if m:
        prefixes = [f"lora_gemm_kernel<DF16_,{m.group(1)},{bn},{1 if m.group(3) == 'true' else 0}" for bn in m.group(2).split("|")]
    else:
        prefixes = [kernel_name.split("<")[0]]
    entries = [v for k, v in table.items() if isinstance(v, dict) and any(k.startswith(p) for p in prefixes)
               and "traffic_bytes_per_launch" in v]
    n = sum(e.get("dispatches", 1) for e in entries)
    return sum(e["traffic_bytes_per_launch"] * e.get("dispatches", 1) for e in entries) / n if n else None


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs but grants a 16-CPU share; oversubscribing the quota stalls every OpenMP region)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(latent, rank_r, steps):
    """The CPU oracle (a restatement of the reference path, kind="port") timed on this box's host cores on a
    bounded sample of the same workload: batch 1 at the same resolution, fp32 (the reference's CPU path)."""
    from harness.unet import UNet2DConditionModel, sd15_config
    from oracle import lora_oracle as orc

    cores = usable_cpus()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host threads, {steps} timed steps + 1 warm-up")
    torch.manual_seed(0)
    unet = UNet2DConditionModel(sd15_config())
    unet.requires_grad_(False)
    params, _ = orc.inject(unet, r=rank_r)
    orc.train_steps(unet, params, 1, 1, latent, 77, 768, lr=1e-4)  # warm-up step (allocator, thread pool)
    t0 = time.perf_counter()
    orc.train_steps(unet, params, steps, 1, latent, 77, 768, lr=1e-4, first_step=1)
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed steps (+1 warm-up) of the same train step at batch 1, {latent}x{latent} latents, "
                      f"fp32, SD1.5-shaped UNet LoRA r={rank_r}, oracle/lora_oracle.py on torch-CPU with {cores} threads",
            "s_per_step": dt / steps}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: become one (before anything in this process has touched the GPU)
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if args.stub_body:
        return stub_body(args, rank, world)
    import torch.distributed as dist

    torch.set_num_threads(max(1, usable_cpus() // max(1, world if world <= 8 else 8)))
    if args.shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from diffusion_finetuning_amd import _native as nat
    from diffusion_finetuning_amd.trainer import LoraTrainer

    dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    unet = build_model(device, dtype, args.rank_r)
    # forward+backward of a step are recorded once into a hipGraph (during the priming step) and replayed; the
    # partial-sum fold, the RCCL exchange and the optimizer are launched from the host every step (trainer.py)
    # (the gloo rehearsal backend stages the slab through the host; with a recorded graph alive in two processes on one
    #  device that path degrades to seconds per step — before and after the recording — so it stays host-launched)
    use_graph = not args.no_graph and (world == 1 or args.backend == "nccl")
    trainer = LoraTrainer(unet, lr=1e-4, capture_graph=use_graph)
    data = synthetic_steps(args.warmup + args.steps, args.batch, args.latent, rank, world, device)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_step(i):
        # the reference draws noise and timesteps inside the step (train_lora_dreambooth.py:824-832): so does the timed step
        # here — on the device, in the prologue kernel (Philox4x32-10 keyed by (seed, optimizer step), rank-invariant);
        # --host-noise feeds the pre-drawn tensors of synthetic_steps instead
        lat, noise, t, ctx = data[i]
        if args.host_noise:
            return trainer.step(lat, noise, t, ctx)
        return trainer.step(lat, None, None, ctx, seed=1000)

    if rank == 0:
        log(f"model + {len(data)} synthetic batches resident on {torch.cuda.get_device_name(local_rank)}; priming")
    # Setup, not measurement: one throw-away step on a scratch copy of the LoRA state so that MIOpen / hipBLASLt /
    # SDPA pick (and, on a box with a cold cache, search for) their kernels before the W warm-up steps start.
    snapshot = (trainer.slab.params.clone(), trainer.opt.exp_avg.clone(), trainer.opt.exp_avg_sq.clone(), trainer.opt.step_count,
                trainer.opt.norm.clone())
    want_graph, trainer.capture_graph = trainer.capture_graph, False
    run_step(0)  # host-launched: solver searches and lazy initialisation happen here
    torch.cuda.synchronize()
    if want_graph:
        # Launch-mode selection, still setup: record the graph, then time two host-launched and two replayed steps.
        # The graph is kept only if it is not slower, and every rank takes the same decision (the two modes issue
        # different collectives), so an unexpected runtime interaction can cost speed but never correctness.
        def trial(n=2):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                run_step(0)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n

        t_host = trial()
        trainer.capture_graph = True
        # records; a rank whose recording fails finishes this step host-launched with the SAME single whole-slab
        # all-reduce a replay issues (LoraTrainer._step_graph), so the collectives stay matched whatever happens
        run_step(0)
        ok = torch.tensor([1.0 if trainer.capture_graph else 0.0], device=device)
        if world > 1:  # agree on the launch mode BEFORE any further step: the two modes bucket the exchange differently
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        trainer.capture_graph = bool(ok.item() > 0)
        t_graph = trial() if trainer.capture_graph else float("inf")
        keep = torch.tensor([1.0 if t_graph <= 1.05 * t_host else 0.0], device=device)
        if world > 1:
            dist.all_reduce(keep, op=dist.ReduceOp.MIN)
        trainer.capture_graph = bool(keep.item() > 0)
        if rank == 0:
            log(f"launch mode: host {1e3 * t_host:.1f} ms/step, hipGraph {1e3 * t_graph:.1f} ms/step -> "
                f"{'hipGraph' if trainer.capture_graph else 'host-launched'}")
    trainer.slab.params.copy_(snapshot[0]); trainer.opt.exp_avg.copy_(snapshot[1]); trainer.opt.exp_avg_sq.copy_(snapshot[2])
    trainer.opt.step_count = snapshot[3]
    trainer.opt.norm.copy_(snapshot[4])  # incl. the device-side count of applied steps
    del snapshot
    if rank == 0:
        log("warm-up")
    losses = []
    for i in range(args.warmup):
        losses.append(run_step(i))
    barrier()
    if rank == 0:
        log(f"timing {args.steps} steps")
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        losses.append(run_step(i))
    barrier()
    elapsed = time.perf_counter() - t0
    # Roofline pass: the SAME K steps again with start/stop events attached to every hot-path dispatch.  It is a
    # second pass because the events serialise consecutive dispatches (≈4 % on the step), which must not leak into
    # `value`; all ranks run it so that the collectives stay matched.
    profiled = not args.no_prof
    prof, elapsed_prof = {}, None
    graph_used = trainer.capture_graph and trainer._graph is not None
    if profiled:
        trainer.capture_graph = False  # events attach to live dispatches: this pass launches from the host
        if rank == 0:
            nat.prof_enable(args.steps * 600)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            run_step(i)
        barrier()
        elapsed_prof = time.perf_counter() - t1
        if rank == 0:
            prof = nat.prof_collect()
            nat.prof_enable(0)
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        log(f"timed region done: {1e3 * elapsed / args.steps:.2f} ms/step")
    final_loss = float(losses[-1].item())
    if rank == 0:
        log("losses: " + " ".join(f"{float(l.item()):.4f}" for l in losses))
    overflow = trainer.opt.overflowed()

    if rank == 0:
        images = world * args.batch * args.steps
        result = {
            "metric": "images/s SD1.5 LoRA rank-4 512^2 train step",
            "value": images / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic latents/text embeddings, random-init SD1.5-shaped UNet (no checkpoints offline)",
            "config": {"workload": f"SD1.5 UNet-only LoRA rank={args.rank_r}, batch={args.batch}/GPU, "
                                   f"{args.latent * 8}^2 ({args.latent}x{args.latent}x4 latents), {args.dtype} storage/compute, "
                                   "fp32 accumulate + fp32 master LoRA, full train step (fwd+bwd+clip+AdamW)",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "lora_params": trainer.slab.numel, "final_loss": final_loss, "overflow": overflow,
                       "hipgraph": bool(graph_used),
                       "noise": "pre-drawn on the host" if args.host_noise else
                                "drawn on the device inside the step (Philox4x32-10 prologue kernel, rank-invariant)"},
        }
        if prof:
            dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
            name, d = dom
            secs = d["ms"] / 1e3
            ai = d["flops"] / d["bytes"]
            ridge = MFMA_PEAK_TFLOPS[args.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
            if ai < ridge:
                roof = {"bound": "hbm", "achieved": d["bytes"] / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            else:
                roof = {"bound": "mfma", "achieved": d["flops"] / secs / 1e12, "peak": MFMA_PEAK_TFLOPS[args.dtype],
                        "unit": "TFLOP/s"}
            roof["frac"] = roof["achieved"] / roof["peak"]
            roof["traffic"] = measured_traffic(name, args.dtype)
            roof.update({"kernel": name, "launches": d["launches"], "avg_us": 1e3 * d["ms"] / d["launches"],
                         "measured": "dispatch-attached HIP events on the launch stream, second pass over the same K steps "
                                     "(events off in the pass that yields `value`)",
                         "ms_per_step_with_events": 1e3 * elapsed_prof / args.steps,
                         "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                         "algorithmic_flops_per_launch": d["flops"] / d["launches"],
                         "hbm_frac": d["bytes"] / secs / 1e9 / HBM_PEAK_GBS,
                         "mfma_frac": d["flops"] / secs / 1e12 / MFMA_PEAK_TFLOPS[args.dtype]})
            result["roofline"] = roof
            hot_ms = sum(v["ms"] for v in prof.values())
            result["hot_path"] = {
                "kernel_ms_per_step": hot_ms / args.steps,
                "share_of_step": hot_ms / args.steps / (1e3 * elapsed_prof / args.steps),
                "kernels": {k: {"launches_per_step": v["launches"] / args.steps, "avg_us": 1e3 * v["ms"] / v["launches"],
                                "GBps": v["bytes"] / v["ms"] / 1e6, "TFLOPs": v["flops"] / v["ms"] / 1e9}
                            for k, v in prof.items()},
            }
        if world == 1 and not args.no_cpu_baseline:
            del trainer, unet, data
            torch.cuda.empty_cache()
            result["cpu_baseline"] = cpu_baseline(args.latent, args.rank_r, args.cpu_steps)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
