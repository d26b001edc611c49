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

# dmabuf IPC (RCCL / device-tensor sharing across the ranks of one node needs it on this driver): set before anything
# initialises the GPU, also when the driver starts the ranks itself with torch.distributed.run
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FLOOR_STEPS = 3  # steps of the launch-floor pass
HBM_ACHIEVABLE_GBS = 6290.0  # "fraction = /8.0 TB/s (also show /6.29)" — the measured streaming ceiling
MFMA_PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}  # dense peaks, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json config to time as the headline (2 = the metric's own; the default run appends 3, 4, 5 "
                         "as `extra_configs`)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (train_batch_size); default: the config's")
    ap.add_argument("--rank-r", type=int, default=None)
    ap.add_argument("--latent", type=int, default=None, help="latent height=width (512² images → 64)")
    ap.add_argument("--no-extra", action="store_true", help="headline only: no extra_configs, no drop-in route")
    ap.add_argument("--drop-in", action="store_true",
                    help="time ONLY the unchanged reference trainer loop (no LoraTrainer): what extra.drop_in reports")
    ap.add_argument("--mask", action="store_true", help="masked loss (cli_lora_pti.py:222-247) on a random binary mask")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--no-conv-autotune", action="store_true",
                    help="caller side: leave MIOpen in immediate mode for the UNet's convolutions (see conv_autotune)")
    ap.add_argument("--no-prof", action="store_true", help="do not attach kernel events in the timed region")
    ap.add_argument("--no-floor", action="store_true", help="skip the launch-floor pass (empty kernels at every hot-path launch site)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel from the host each step instead of replaying the recorded hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-noise", action="store_true",
                    help="feed pre-drawn noise / timesteps instead of drawing them on the device inside the step")
    ap.add_argument("--cpu-steps", type=int, default=3,
                    help="timed oracle steps of cpu_baseline.same_resolution (~3 s each on 16 cores; cfg1 always runs its 10)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --shared-gpu rehearses N ranks on one GPU")
    ap.add_argument("--shared-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--exchange-probe", action="store_true", help="only the 1-rank RCCL exchange probe (extra.exchange_probe)")
    ap.add_argument("--stub-body", default=None, choices=["ok", "fail", "diverge"],
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


def replicas_identical(flat, dist, rank) -> bool:
    """Two-word signature of a rank's trainable state (sum and index-weighted sum, float64), MIN- and MAX-reduced over the
    ranks: equal on every rank, or some replica diverged (a missed collective, a different loss scale)."""
    chk = flat.double()
    sig = torch.stack([chk.sum(), (chk * torch.arange(1, chk.numel() + 1, device=chk.device, dtype=torch.float64)).sum()])
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        log(f"rank {rank}: LoRA slab differs across ranks after the rehearsal steps ({sig.tolist()} vs min {lo.tolist()} / "
            f"max {hi.tolist()}) — aborting")
        return False
    return True


def stub_body(args, rank, world):
    """Launcher rehearsal body (CPU only, gloo): the data-parallel skeleton of the real body without a GPU — every rank holds a
    replica of a small slab, takes `steps` steps on its own shard with the REAL exchange (trainer.SlabExchange: one SUM
    all-reduce of the gradient slab, 1/world folded into the update), then the ranks compare replica signatures exactly as
    the real body does (exit code 4 on a divergence); rank 0 prints ONE line with the contract's keys, `config.rccl_ranks` /
    `backend` as the process group reports them.  'fail' makes rank 1 exit non-zero (error propagation of the launcher);
    'diverge' makes the last rank skip the exchange of one step."""
    import torch.distributed as dist

    from diffusion_finetuning_amd.trainer import SlabExchange

    if args.stub_body == "fail" and rank == 1:
        sys.exit(3)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    g = torch.Generator().manual_seed(0)
    slab = torch.randn(1024, generator=g)            # identical initial replicas (the real body broadcasts rank 0's)
    grads = torch.zeros_like(slab)
    ex = SlabExchange(grads, grads.numel(), None)
    ex.single = True                                 # a recorded step's exchange: the whole slab in one all-reduce
    for step in range(args.warmup + args.steps):
        gs = torch.Generator().manual_seed(1000 * step + rank)   # every rank its own shard
        grads.copy_(torch.randn(1024, generator=gs))
        ex.arm()
        if not (args.stub_body == "diverge" and rank == world - 1 and step == 1):
            ex.finish()
        elif world > 1:  # keep the collective sequence (nobody hangs), but train on something else
            dist.all_reduce(grads.clone())
        slab -= 1e-2 * grads / world
    if world > 1 and not replicas_identical(slab, dist, rank):
        sys.exit(4)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": float(t.item()), "unit": "ranksum", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "stub": True, "scaling": "weak",
                          "config": {"parallelism": f"dp{world}", "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                                     "backend": dist.get_backend() if world > 1 else None}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# BASELINE.json `configs` (index = position in that list; 1 is the CPU reference case and is cpu_baseline's workload).
# `batch` = images (instance rows) per GPU per step; cfg-4 adds as many class rows (prior preservation,
# train_lora_dreambooth.py:698-702), so its UNet sees 2·batch rows.
CONFIGS = {
    2: dict(tag="cfg-2", unet="sd15", rank=4, batch=4, latent=64, ctx_len=77, ctx_dim=768, prior=False, text_encoder=False,
            v_prediction=False, what="SD1.5 UNet-only LoRA rank={rank}, batch={batch}/GPU"),
    3: dict(tag="cfg-3", unet="sd15", rank=8, batch=4, latent=64, ctx_len=77, ctx_dim=768, prior=False, text_encoder=True,
            v_prediction=False, what="SD1.5 UNet + CLIP-L text-encoder LoRA rank={rank} (--train_text_encoder: one --lora_rank "
                                     "for both, train_lora_dreambooth.py:596-613), batch={batch}/GPU"),
    4: dict(tag="cfg-4", unet="sd15", rank=4, batch=4, latent=64, ctx_len=77, ctx_dim=768, prior=True, text_encoder=False,
            v_prediction=False, what="SD1.5 Dreambooth LoRA rank={rank} with prior preservation, {batch} instance + {batch} "
                                     "class rows per GPU (global batch 32 = 4 x 8 GPUs)"),
    # cli_lora_pti.py's tuning phase as it runs by default (continue_inversion=True, :528): UNet LoRA + the text encoder's
    # token-embedding table in one AdamW (:706-738, weight_decay_lora 1e-3), the encoder run inside the step (:199-206),
    # timesteps below int(1000·0.8) (:444), v-prediction (SD2.1-768)
    5: dict(tag="cfg-5", unet="sd21-768", rank=16, batch=1, latent=96, ctx_len=77, ctx_dim=1024, prior=False,
            text_encoder="openclip-h-ti", v_prediction=True, t_multiplier=0.8, weight_decay=1e-3,
            what="SD2.1-768 PTI tuning step (cli_lora_pti perform_tuning, continue_inversion): UNet LoRA rank={rank} + the "
                 "trainable token-embedding table (49408x1024, 'extended-latent TI') of an OpenCLIP-H-shaped text encoder run "
                 "inside the step, attn2 to_k/to_v with dX, t < 800, v-prediction, batch={batch}/GPU"),
}


def build_unet(device, dtype, rank_r, kind="sd15", hooks=True):
    import diffusion_finetuning_amd as dfa
    from harness.unet import UNet2DConditionModel, sd15_config, sd21_768_config

    torch.manual_seed(0)  # identical random-init weights on every rank (no checkpoints offline)
    with torch.device(device):
        unet = UNet2DConditionModel(sd15_config() if kind == "sd15" else sd21_768_config())
    unet = unet.to(dtype)
    unet.requires_grad_(False)  # train_lora_dreambooth.py:595
    dfa.inject_trainable_lora(unet, r=rank_r)  # :596-598
    if hooks:
        from diffusion_finetuning_amd.attention import set_use_hip_geglu, set_use_memory_efficient_attention_xformers

        set_use_memory_efficient_attention_xformers(unet, True)  # :623-624 (--use_xformers): here the HIP attention cores
        set_use_hip_geglu(unet, True)  # the fused GEGLU gate after each `proj` LoRA linear (idempotent)
    g = torch.Generator(device="cpu").manual_seed(1)
    with torch.no_grad():  # warm-started `up` so no kernel sees the all-zero branch
        for up, _ in dfa.extract_lora_ups_down(unet):
            up.weight.copy_((torch.randn(up.weight.shape, generator=g) * 0.01).to(device))
    return unet


def build_model(device, dtype, rank_r):  # (tools/ use this name)
    return build_unet(device, dtype, rank_r)


def build_text_encoder(device, dtype, rank_r, kind=True):
    """kind True / "clip-l-lora": CLIP-L-shaped text encoder (hidden 768, 12 layers, 12 heads, MLP 3072, 77 positions — the
    SD1.5 text encoder's config), random init, LoRA on its CLIPAttention projections (lora.py:54,
    train_lora_dreambooth.py:608-621).  "openclip-h-ti": OpenCLIP-H-shaped (hidden 1024, 23 layers, 16 heads, MLP 4096 — the
    SD2.1 text encoder's config), frozen except its token-embedding table (cli_lora_pti.py:704-722, continue_inversion)."""
    import diffusion_finetuning_amd as dfa
    from transformers import CLIPTextConfig, CLIPTextModel

    torch.manual_seed(2)
    if kind == "openclip-h-ti":
        cfg = CLIPTextConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                             vocab_size=49408, max_position_embeddings=77, bos_token_id=49406, eos_token_id=49407,
                             pad_token_id=0, hidden_act="gelu")
        te = CLIPTextModel(cfg)
        te.requires_grad_(False)
        te = te.to(device).to(dtype)
        te.get_input_embeddings().weight.requires_grad_(True)  # the one trainable tensor of the encoder (:708-722)
        return te
    cfg = CLIPTextConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                         vocab_size=49408, max_position_embeddings=77, bos_token_id=49406, eos_token_id=49407, pad_token_id=1)
    te = CLIPTextModel(cfg)
    te.requires_grad_(False)
    te = te.to(device).to(dtype)
    dfa.inject_trainable_lora(te, target_replace_module=dfa.TEXT_ENCODER_DEFAULT_TARGET_REPLACE, r=rank_r)
    g = torch.Generator(device="cpu").manual_seed(3)
    with torch.no_grad():
        for up, _ in dfa.extract_lora_ups_down(te, target_replace_module=dfa.TEXT_ENCODER_DEFAULT_TARGET_REPLACE):
            up.weight.copy_((torch.randn(up.weight.shape, generator=g) * 0.01).to(device))
    return te


def synthetic_steps(n_steps, batch, latent, rank, world, device, ctx_len=77, ctx_dim=768, rows_per_image=1, ids=False):
    """Per-step inputs, resident in HBM: noise/timesteps are rank-invariant (set_seed semantics,
    train_lora_dreambooth.py:509-510); each rank owns its shard of the latents / text embeddings (or token ids).
    rows_per_image = 2 under prior preservation: `batch` instance rows then `batch` class rows (:698-702)."""
    out = []
    rows = batch * rows_per_image
    for s in range(n_steps):
        g = torch.Generator().manual_seed(1000 + s)
        lat = torch.randn(world * rows, 4, latent, latent, generator=g) * 0.18215
        ctx = torch.randn(world * rows, ctx_len, ctx_dim, generator=g)
        noise = torch.randn(rows, 4, latent, latent, generator=g)
        t = torch.randint(0, 1000, (rows,), generator=g)
        tok = torch.randint(2, 49000, (world * rows, ctx_len), generator=g)
        tok[:, 0], tok[:, 24:] = 49406, 49407  # caption-shaped: bos, 23 words, eos repeated to the end (tokenizer padding)
        sl = slice(rank * rows, (rank + 1) * rows)
        cond = tok[sl].to(device) if ids else ctx[sl].to(device)
        out.append((lat[sl].to(device), noise.to(device), t.to(device), cond))
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
    # (",g2" = the gated frozen ff.net.2 backward GEMM and "splitk" = the split-K instantiations: kinds of their own)
    entries = [v for k, v in table.items() if isinstance(v, dict) and any(k.startswith(p) for p in prefixes)
               and ",g2" not in k and "splitk" not in k and "traffic_bytes_per_launch" in v]
    n = sum(e.get("dispatches", 1) for e in entries)
    return sum(e["traffic_bytes_per_launch"] * e.get("dispatches", 1) for e in entries) / n if n else None


MIOPEN_DB = os.path.join(ROOT, "harness", "miopen_db")


def conv_autotune(args) -> bool:
    """Caller side, not the hot path: the UNet's convolutions (stock MIOpen, ≈ 10 ms of the step) run with MIOpen's solver SEARCH
    on (`torch.backends.cudnn.benchmark`) instead of its immediate-mode heuristics — +1–3 % images/s on the headline.  A search
    costs ≈ 3 minutes per workload on a fresh box, so its RESULTS ship with the harness: `harness/miopen_db/*.ufdb.txt` is MIOpen's
    own user find-db (text, keyed by problem and gfx950 / 256 CUs / MIOpen version), written by `BENCH_MIOPEN_DB_INPLACE=1 bench.py`
    runs on the GPU box and handed to MIOpen through MIOPEN_USER_DB_PATH (a private per-process copy); with it the priming step finds every solver without benchmarking.  If the db is absent, or
    a priming step shows that a search is running after all (run_workload), the setting is dropped."""
    import glob

    if args.no_conv_autotune or os.environ.get("BENCH_CONV_AUTOTUNE", "1") == "0" or not glob.glob(os.path.join(MIOPEN_DB, "*.ufdb.txt")):
        return False
    if "MIOPEN_USER_DB_PATH" not in os.environ:
        if os.environ.get("BENCH_MIOPEN_DB_INPLACE", "0") == "1":  # how the shipped records were written: the search's results land in the tree
            path = MIOPEN_DB
        else:
            # a private copy per process (two small text files): N ranks do not queue on MIOpen's lock files, and a box whose MIOpen
            # adds a record does not edit tracked files.  The drop-in child process inherits the variable and reads this copy.
            import atexit
            import shutil
            import tempfile

            path = tempfile.mkdtemp(prefix="dfa_miopen_db_")
            for f in glob.glob(os.path.join(MIOPEN_DB, "*.txt")):
                shutil.copy(f, path)
            atexit.register(shutil.rmtree, path, ignore_errors=True)
        os.environ["MIOPEN_USER_DB_PATH"] = path
    torch.backends.cudnn.benchmark = True
    return True


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


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(rank_r, same_res_latent, same_res_steps):
    """The CPU oracle (a restatement of the reference path, kind="port") timed on this box's host cores.
    `cfg1`: the workload BASELINE.md §5 defines — BASELINE config 1, SD1.5-shaped UNet LoRA r=4, batch 1, 256² (32×32×4
    latents), fp32, 10 steps of the reference's loop (train_lora_dreambooth.py:811-888) — which is also the parity target of
    tests/test_gpu_parity.py::test_cfg1_full_size_sd15_fp32_trajectory_vs_cpu_oracle.  `same_resolution`: a bounded sample of
    the GPU headline's own resolution (batch 1).  Neither is the target; the roofline fraction is."""
    from harness.unet import UNet2DConditionModel, sd15_config
    from oracle import lora_oracle as orc

    cores = usable_cpus()
    torch.set_num_threads(cores)
    out = {"unit": "images/s", "cores": cores, "cpu_model": cpu_model_name(), "kind": "port"}

    def run(latent, steps):
        torch.manual_seed(0)
        unet = UNet2DConditionModel(sd15_config())
        unet.requires_grad_(False)
        params, _ = orc.inject(unet, r=rank_r)
        orc.train_steps(unet, params, 1, 1, latent, 77, 768, lr=1e-4)  # warm-up step (allocator, thread pool)
        t0 = time.perf_counter()
        orc.train_steps(unet, params, steps, 1, latent, 77, 768, lr=1e-4, first_step=1)
        return (time.perf_counter() - t0) / steps

    log(f"cpu_baseline: oracle on {cores} host threads ({out['cpu_model']}): cfg-1, 10 timed steps + 1 warm-up")
    s1 = run(32, 10)
    out["cfg1"] = {"images_s": 1.0 / s1, "s_per_step": s1, "steps": 10,
                   "workload": "BASELINE config 1: SD1.5 UNet-only LoRA r=4, batch 1, 256^2 (32x32x4 latents), fp32"}
    log(f"cpu_baseline: {same_res_steps} steps at {same_res_latent}x{same_res_latent} latents")
    s2 = run(same_res_latent, same_res_steps)
    out["same_resolution"] = {"images_s": 1.0 / s2, "s_per_step": s2, "steps": same_res_steps,
                              "workload": f"the headline's resolution at batch 1 ({same_res_latent}x{same_res_latent}x4 latents), fp32"}
    out["value"] = out["cfg1"]["images_s"]
    out["s_per_step"] = s1
    out["sample"] = (f"cfg-1 as BASELINE.md §5 defines it: 10 timed steps (+1 warm-up) of the reference train step at batch 1, "
                     f"32x32 latents (256^2), fp32, SD1.5-shaped UNet LoRA r={rank_r}, oracle/lora_oracle.py on torch-CPU with "
                     f"{cores} threads of {out['cpu_model']}; `same_resolution` = {same_res_steps} steps at the headline's 512^2")
    return out


# hot-path kernel kinds that are SURVEY §8(d) LoRA layers (fwd / dX GEMMs, P-only launches, factor gradients, loss) — the
# roofline object is built from these; the gated frozen GEMM (ff.net.2 backward + GEGLU gate) and the attention cores
# (§8 f-4) are reported next to them, never mixed into the LoRA classes
def add_launch_floor(roof, floor, tot_ms):
    """Launch floor into `roof["step_level"]` and the per-class table.  `floor`: per profiler
    kind {"launches", "ms"} PER STEP of the pass in which every hot-path launch site dispatched an EMPTY kernel of the same
    grid / block / LDS / argument segment, in the step's own order; `tot_ms`: §8(d) kernel time per step of the real pass.
    `launch_floor_ms` = Σ over the §8(d) kinds; `frac_vs_launch_bounded` = HBM bound ÷ (kernel time − floor): what per-layer
    launches could reach at best — `frac` stays the contract's figure against ALL kernel time."""
    fl_lora = {k: v for k, v in floor.items() if _is_lora_kind(k)}
    if not fl_lora:
        return
    floor_ms = sum(v["ms"] for v in fl_lora.values())
    sl = roof["step_level"]
    sl["launch_floor_ms"] = floor_ms
    sl["launch_floor_launches_per_step"] = sum(v["launches"] for v in fl_lora.values())
    sl["launch_floor_us_per_launch"] = 1e3 * floor_ms / max(sl["launch_floor_launches_per_step"], 1)
    sl["frac_vs_launch_bounded"] = sl["hbm_bound_ms"] / max(tot_ms - floor_ms, 1e-9)
    # what `frac` could be at best with ONE launch per layer / group: every kernel at the HBM peak, the floors still paid
    sl["frac_ceiling_with_per_layer_launches"] = sl["hbm_bound_ms"] / (sl["hbm_bound_ms"] + floor_ms)
    sl["launch_floor_share_of_kernel_time"] = floor_ms / tot_ms
    sl["launch_floor_by_kind_us"] = {k: 1e3 * v["ms"] / v["launches"] for k, v in fl_lora.items()}
    for k, c in roof.get("fused_gemm_classes", {}).items():
        if k in fl_lora:
            c["launch_floor_us_per_launch"] = 1e3 * fl_lora[k]["ms"] / fl_lora[k]["launches"]
            c["us_per_launch"] = 1e3 * c["ms_per_step"] / c["launches_per_step"]
    # (the other ablation of DESIGN.md §5 — the K-loop's loads AND multiplications switched off, LORA_GEMM_DBG=3, a diagnostic
    # build whose results are wrong — is not run here: its per-class figures sit in profiles/README.md)
    sl["launch_floor_all_library_kinds_ms"] = sum(v["ms"] for v in floor.values())


def _is_lora_kind(name):
    return name.startswith("lora_") or name.startswith("ddpm_")


def run_workload(args, cfg_id, rank, world, device, dist, profile=True):
    """Build the workload of BASELINE config `cfg_id`, time args.steps steps of it, optionally run the event pass."""
    from diffusion_finetuning_amd import _native as nat
    from diffusion_finetuning_amd.trainer import LoraTrainer

    cfg = dict(CONFIGS[cfg_id])
    if cfg_id == args.config:  # command-line overrides apply to the headline workload only
        cfg["batch"] = args.batch if args.batch is not None else cfg["batch"]
        cfg["rank"] = args.rank_r if args.rank_r is not None else cfg["rank"]
        cfg["latent"] = args.latent if args.latent is not None else cfg["latent"]
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    unet = build_unet(device, dtype, cfg["rank"], cfg["unet"])
    te = build_text_encoder(device, dtype, cfg["rank"], cfg["text_encoder"]) if cfg["text_encoder"] else None
    # forward+backward(+factor gradients) of a step are recorded once into a hipGraph (during the priming step) and
    # replayed; the RCCL exchange and the optimizer are launched from the host every step (trainer.py)
    # (the gloo rehearsal backend stages the slab through the host; with a recorded graph alive in two processes on one
    #  device that path degrades to seconds per step — before and after the recording — so it stays host-launched)
    use_graph = not args.no_graph and (world == 1 or args.backend == "nccl")
    trainer = LoraTrainer(unet, te, lr=1e-4, lr_text=5e-5, lr_embed=5e-4, weight_decay=cfg.get("weight_decay", 1e-2),
                          capture_graph=use_graph, v_prediction=cfg["v_prediction"])
    rows_per_image = 2 if cfg["prior"] else 1
    data = synthetic_steps(args.warmup + args.steps, cfg["batch"], cfg["latent"], rank, world, device, cfg["ctx_len"],
                           cfg["ctx_dim"], rows_per_image, ids=te is not None)
    mask = None
    if args.mask:
        g = torch.Generator().manual_seed(5)
        mask = (torch.rand(cfg["batch"] * rows_per_image, 1, cfg["latent"] * 8, cfg["latent"] * 8, generator=g) > 0.5).float().to(device)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_step(i):
        # the reference draws noise and timesteps inside the step (train_lora_dreambooth.py:824-832): so does the timed step
        # here — on the device, in the prologue kernel (Philox4x32-10 keyed by (seed, optimizer step), rank-invariant);
        # --host-noise feeds the pre-drawn tensors of synthetic_steps instead
        lat, noise, t, cond = data[i]
        kw = dict(with_prior_preservation=cfg["prior"], mask=mask, t_multiplier=cfg.get("t_multiplier", 1.0))
        kw["input_ids" if te is not None else "encoder_hidden_states"] = cond
        if args.host_noise:
            return trainer.step(lat, noise, t, **kw)
        return trainer.step(lat, None, None, seed=1000, **kw)

    if rank == 0:
        log(f"[{cfg['tag']}] model + {len(data)} synthetic batches resident on {torch.cuda.get_device_name(device)}; priming")
    # Setup, not measurement: one throw-away step on a scratch copy of the LoRA state so that MIOpen / hipBLASLt /
    # SDPA pick (and, on a box with a cold cache, search for) their kernels before the W warm-up steps start.
    snapshot = (trainer.slab.params.clone(), trainer.opt.exp_avg.clone(), trainer.opt.exp_avg_sq.clone(), trainer.opt.step_count,
                trainer.opt.norm.clone())
    want_graph, trainer.capture_graph = trainer.capture_graph, False
    t_prime = time.perf_counter()
    run_step(0)  # host-launched: solver searches and lazy initialisation happen here
    torch.cuda.synchronize()
    t_prime = time.perf_counter() - t_prime
    if torch.backends.cudnn.benchmark and t_prime > 60.0:
        # the shipped find-db did not cover this box (another MIOpen build?): MIOpen searched.  Keep what it found for this
        # workload, but do not pay a search for every further workload of the run
        log(f"[{cfg['tag']}] priming took {t_prime:.0f} s: MIOpen searched its solvers (find-db miss) — immediate mode from here on")
        torch.backends.cudnn.benchmark = False
        os.environ["BENCH_CONV_AUTOTUNE"] = "0"  # (inherited by the drop-in child process)
    launch_trial = None
    if want_graph:
        # Launch-mode selection, still setup: record the graph, then time TRIAL host-launched and TRIAL replayed steps one by
        # one (after one discarded step of each mode) and compare the MEDIANS — two-step means were decided by noise (round 3:
        # the driver box and the builder's box chose differently for cfg-4).  The graph is kept unless its median loses by more
        # than 5 %; every rank takes the same decision (MAX of the medians over ranks).  The collectives are the same in both
        # modes — one whole-slab all-reduce per step once a recording was asked for, trainer.py — so the choice is about speed.
        TRIAL = 5

        def trial():
            run_step(0)  # discarded: the first step of a mode pays its one-off costs (allocator growth, the first replay)
            times = []
            for _ in range(TRIAL):
                torch.cuda.synchronize()
                t = time.perf_counter()
                run_step(0)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t)
            med = torch.tensor([sorted(times)[TRIAL // 2]], device=device, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(med, op=dist.ReduceOp.MAX)
            return float(med.item())

        t_host = trial()
        trainer.capture_graph = True
        run_step(0)  # records (a failed recording finishes the step host-launched and clears trainer.capture_graph)
        ok = torch.tensor([1.0 if trainer.capture_graph else 0.0], device=device)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        trainer.capture_graph = bool(ok.item() > 0)
        t_graph = trial() if trainer.capture_graph else float("inf")
        trainer.capture_graph = t_graph <= 1.05 * t_host
        launch_trial = {"host_median_ms": 1e3 * t_host, "hipgraph_median_ms": None if t_graph == float("inf") else 1e3 * t_graph,
                        "steps_per_mode": TRIAL, "rule": "keep the recorded step unless its median loses by > 5 %",
                        "chosen": "hipGraph" if trainer.capture_graph else "host-launched"}
        if rank == 0:
            log(f"[{cfg['tag']}] launch mode (median of {TRIAL}): host {1e3 * t_host:.2f} ms/step, hipGraph {1e3 * t_graph:.2f} "
                f"ms/step -> {launch_trial['chosen']}")
    if world > 1:
        # Replicas must still be identical after steps in BOTH launch modes (each rank trained on its own shard): a rank
        # that diverged — a missed collective, a different loss scale — is a broken run, not a slow one.
        graph_mode = trainer.capture_graph
        for mode in (False, graph_mode):
            trainer.capture_graph = mode
            run_step(0)
        if not replicas_identical(trainer.slab.params, dist, rank):  # LoRA region AND the dense tail (config 5: the token table)
            sys.exit(4)
    trainer.slab.params.copy_(snapshot[0]); trainer.opt.exp_avg.copy_(snapshot[1]); trainer.opt.exp_avg_sq.copy_(snapshot[2])
    trainer.opt.step_count = snapshot[3]
    trainer.opt.norm.copy_(snapshot[4])  # incl. the device-side count of applied steps
    del snapshot
    if rank == 0:
        log(f"[{cfg['tag']}] warm-up")
    losses = []
    for i in range(args.warmup):
        losses.append(run_step(i))
    barrier()
    if rank == 0:
        log(f"[{cfg['tag']}] timing {args.steps} steps")
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        losses.append(run_step(i))
    barrier()
    elapsed = time.perf_counter() - t0
    # What a step does OUTSIDE the recording, per rank: [RCCL all-reduce of the slab] → clip norm + AdamW → re-pack of the
    # compute-dtype factors (+ the token-table row sum of config 5).  Timed with events in a few extra replayed steps after the
    # timed region (not part of `value`); on one GPU the exchange is empty, so this is the floor the 8-GPU step adds its
    # all-reduce to.
    tail_ms = None
    if trainer.capture_graph and trainer._graph is not None:
        trainer.tail_events = []
        for i in range(args.warmup, args.warmup + min(5, args.steps)):
            run_step(i)
        torch.cuda.synchronize()
        tail_ms = sorted(a.elapsed_time(b) for a, b in trainer.tail_events)[len(trainer.tail_events) // 2]
        trainer.tail_events = None
    # Roofline pass: the SAME K steps again with start/stop events attached to every hot-path dispatch.  It is a
    # second pass because the events serialise consecutive dispatches (≈4 % on the step), which must not leak into
    # `value`; all ranks run it so that the collectives stay matched.
    prof, elapsed_prof = {}, None
    graph_used = trainer.capture_graph and trainer._graph is not None
    if profile:
        trainer.capture_graph = False  # events attach to live dispatches: this pass launches from the host
        if rank == 0:
            nat.prof_enable(args.steps * 900)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            run_step(i)
        barrier()
        elapsed_prof = time.perf_counter() - t1
        if rank == 0:
            prof = nat.prof_collect()
            nat.prof_enable(0)
    overflow = trainer.opt.overflowed()
    # Launch-floor pass (single GPU, headline only): the same host-launched step once more with every hot-path
    # launch site dispatching an EMPTY kernel of the same grid, block, LDS and argument segment (lora_prof_null_mode), in the
    # step's own order between the caller's real kernels.  What the events record is what a launch of that shape costs before it
    # does any work: dispatch, wave launch ramp, argument fetch, drain.  Nothing is computed in this pass (outputs are not
    # written), so it runs LAST: the trainer is discarded right after.
    floor = {}
    if profile and world == 1 and cfg_id == args.config and not args.no_floor:
        nat.prof_null_mode(True)
        try:
            nat.prof_enable(FLOOR_STEPS * 900)
            for i in range(args.warmup, args.warmup + FLOOR_STEPS):
                run_step(i)
            torch.cuda.synchronize()
            floor = {k: {"launches": v["launches"] / FLOOR_STEPS, "ms": v["ms"] / FLOOR_STEPS} for k, v in nat.prof_collect().items()}
        finally:
            nat.prof_enable(0)
            nat.prof_null_mode(False)
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_loss = float(losses[-1].item())
    if rank == 0:
        log(f"[{cfg['tag']}] timed region done: {1e3 * elapsed / args.steps:.2f} ms/step; losses: " +
            " ".join(f"{float(l.item()):.4f}" for l in losses))
    res = {"cfg": cfg, "elapsed": elapsed, "elapsed_prof": elapsed_prof, "prof": prof, "graph_used": bool(graph_used),
           "tail_ms": tail_ms, "launch_trial": launch_trial, "survey": trainer.slab.survey_work(2 if args.dtype != "f32" else 4, contract=True),
           "survey_as_run": trainer.slab.survey_work(2 if args.dtype != "f32" else 4),
           "final_loss": final_loss, "overflow": overflow, "lora_params": trainer.slab.numel, "launch_floor": floor,
           "rows_per_image": rows_per_image}
    del trainer, unet, te, data
    torch.cuda.empty_cache()
    return res


def exchange_probe(args, device):
    """VERDICT r4 N1 — the exchange against the compute around it, on ONE GPU through a 1-rank RCCL group (`always_reduce`: the
    all-reduce calls, their stream hand-offs and their kernels are real; no bytes cross a link).  Three forms of the config-2
    step, each on a fresh model: (a) the default — recorded step, ONE whole-slab all-reduce behind the replay; (b) host-launched,
    one all-reduce after backward; (c) host-launched with `early_bucket=True` — the context K/V group cut by block range, the
    [up|mid] bucket all-reduced from the mid block's backward hook while the down blocks still run backward (DDP's buckets fire
    inside backward: train_lora_dreambooth.py:744-757,877).  Reported per form: ms/step (median of 7) and the device time of the
    step's TAIL, from the end of backward to the re-packed factors (what the exchange adds to is in there)."""
    import torch.distributed as dist

    from diffusion_finetuning_amd.trainer import LoraTrainer

    if dist.is_initialized():
        return {"skipped": "a process group is already alive"}
    cfg = CONFIGS[2]
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=device)
    out = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "slab_MB": None,
           "note": "1-rank RCCL group on one GPU: collective calls and kernels are real, nothing crosses xGMI"}
    try:
        data = synthetic_steps(3, cfg["batch"], cfg["latent"], 0, 1, device, cfg["ctx_len"], cfg["ctx_dim"], 1)
        forms = (("recorded_single_allreduce", True, False), ("host_single_allreduce", False, False),
                 ("host_early_bucket", False, True))
        for name, graph, early in forms:
            unet = build_unet(device, dtype, cfg["rank"], cfg["unet"])
            trainer = LoraTrainer(unet, None, lr=1e-4, capture_graph=graph, always_reduce=True, early_bucket=early)
            assert trainer.exchange.active
            out["slab_MB"] = trainer.slab.numel * 4 / 1e6

            def step(i):
                lat, _, _, cond = data[i % len(data)]
                return trainer.step(lat, None, None, seed=1000, encoder_hidden_states=cond)

            for i in range(4):  # priming, recording, first replay
                step(i)
            times = []
            for i in range(7):
                torch.cuda.synchronize()
                t = time.perf_counter()
                step(i)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t)
            trainer.tail_events = []
            for i in range(5):
                step(i)
            torch.cuda.synchronize()
            tails = sorted(a.elapsed_time(b) for a, b in trainer.tail_events)
            trainer.tail_events = None
            out[name] = {"ms_per_step": 1e3 * sorted(times)[len(times) // 2], "tail_ms_per_step": tails[len(tails) // 2],
                         "collectives_per_step": 1 if not early else 2, "ctx_kv_groups": len(trainer.slab.ctx_groups),
                         "replayed": bool(trainer._graph is not None)}
            log(f"exchange probe [{name}]: {out[name]}")
            del trainer, unet
            torch.cuda.empty_cache()
    finally:
        dist.destroy_process_group()
    return out


def hot_path_summary(prof, steps, elapsed_prof):
    lora_ms = sum(v["ms"] for k, v in prof.items() if _is_lora_kind(k))
    all_ms = sum(v["ms"] for v in prof.values())
    return {
        "kernel_ms_per_step": lora_ms / steps,                      # SURVEY §8(a)/(d) kernels: LoRA GEMMs, gradients, loss
        "kernel_ms_per_step_incl_f4": all_ms / steps,               # + gated frozen GEMM + attention cores (§8 f-4)
        "share_of_step": all_ms / steps / (1e3 * elapsed_prof / steps),
        "kernels": {k: {"launches_per_step": v["launches"] / steps, "avg_us": 1e3 * v["ms"] / v["launches"],
                        "ms_per_step": v["ms"] / steps, "GBps": v["bytes"] / v["ms"] / 1e6, "TFLOPs": v["flops"] / v["ms"] / 1e9,
                        "hbm_frac": v["bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS}
                    for k, v in prof.items()},
    }


def drop_in_route(args, device):
    """What an UNCHANGED reference trainer gets (train_lora_dreambooth.py:595-598,623-625,659-676,811-888 under
    `--mixed_precision fp16 --use_xformers`): fp32 module under autocast, inject_trainable_lora, the attention switch,
    itertools.chain of the returned generators into torch.optim.AdamW, F.mse_loss on .float(), GradScaler (what
    accelerator.backward / clip_grad_norm_ drive under fp16), clip_grad_norm_, zero_grad.  No LoraTrainer, no slab, no graph."""
    import itertools

    import diffusion_finetuning_amd as dfa
    import torch.nn.functional as F
    from diffusion_finetuning_amd.attention import set_use_memory_efficient_attention_xformers
    from diffusion_finetuning_amd.trainer import ddpm_tables
    from harness.unet import UNet2DConditionModel, sd15_config

    cfg = CONFIGS[2]
    torch.manual_seed(0)
    with torch.device(device):
        unet = UNet2DConditionModel(sd15_config())
    unet.requires_grad_(False)
    params, _ = dfa.inject_trainable_lora(unet, r=cfg["rank"])
    set_use_memory_efficient_attention_xformers(unet, True)
    plist = list(itertools.chain(*params))
    opt = torch.optim.AdamW(plist, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    sa, sb = ddpm_tables(device=device)
    data = synthetic_steps(args.warmup + args.steps, cfg["batch"], cfg["latent"], 0, 1, device)

    def step(i):
        lat, _, _, ctx = data[i]
        noise = torch.randn_like(lat)                                   # :824
        t = torch.randint(0, 1000, (lat.shape[0],), device=device)      # :826-832
        noisy = sa[t].view(-1, 1, 1, 1) * lat + sb[t].view(-1, 1, 1, 1) * noise  # scheduler.add_noise :837
        with torch.autocast("cuda", dtype=torch.float16):
            pred = unet(noisy, t, ctx).sample                           # :843
        loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")  # :875
        scaler.scale(loss).backward()                                   # accelerator.backward :877
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(plist, 1.0)                      # :878-884
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()                                                 # :888
        return loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out = {"images_s": cfg["batch"] / dt, "ms_per_step": 1e3 * dt, "final_loss": float(loss.item()),
           "route": "unchanged reference trainer loop: fp32 module + autocast(f16), inject_trainable_lora, "
                    "set_use_memory_efficient_attention_xformers, torch.optim.AdamW + GradScaler + clip_grad_norm_, host-launched"}
    del unet, opt, data
    torch.cuda.empty_cache()
    return out


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

    conv_autotune(args)
    torch.set_num_threads(max(1, usable_cpus() // max(1, world if world <= 8 else 8)))
    if args.shared_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        # one process per GPU: refuse loudly rather than let two ranks time-slice one device and report it as N GPUs
        log(f"rank {rank}: {torch.cuda.device_count()} visible GPU(s) for {world} ranks on this node; pass --shared-gpu "
            "for a functional rehearsal on one device (its numbers mean nothing)")
        sys.exit(5)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    if args.exchange_probe:
        print(json.dumps({"metric": "exchange probe", "n_gpus": 1, "extra": {"exchange_probe": exchange_probe(args, device)}}), flush=True)
        return
    if args.drop_in:  # only the unchanged-trainer route
        if rank == 0:
            print(json.dumps({"metric": "images/s SD1.5 LoRA rank-4 512^2 train step, unchanged reference trainer loop",
                              "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                              **{("value" if k == "images_s" else k): v for k, v in drop_in_route(args, device).items()}}), flush=True)
        return

    head = run_workload(args, args.config, rank, world, device, dist, profile=not args.no_prof)
    cfg, elapsed, prof, elapsed_prof = head["cfg"], head["elapsed"], head["prof"], head["elapsed_prof"]

    if rank == 0:
        images = world * cfg["batch"] * args.steps
        what = cfg["what"].format(**cfg)
        result = {
            "metric": "images/s SD1.5 LoRA rank-4 512^2 train step" if args.config == 2 else f"images/s {cfg['tag']} train step",
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
            "data": "synthetic latents/text embeddings, random-init SD-shaped UNet (no checkpoints offline)",
            "config": {"workload": f"{what}, {cfg['latent'] * 8}^2 ({cfg['latent']}x{cfg['latent']}x4 latents), "
                                   f"{args.dtype} storage/compute, fp32 accumulate + fp32 master LoRA, full train step "
                                   "(fwd+bwd+clip+AdamW)",
                       "baseline_config": args.config,
                       "global_batch": world * cfg["batch"], "rows_per_step_per_gpu": cfg["batch"] * head["rows_per_image"],
                       "parallelism": f"dp{world}",
                       "lora_params": head["lora_params"], "final_loss": head["final_loss"], "overflow": head["overflow"],
                       "hipgraph": head["graph_used"], "launch_mode_trial": head["launch_trial"],
                       # caller side: MIOpen solver search for the UNet's convolutions, results shipped in harness/miopen_db
                       "conv_autotune": bool(torch.backends.cudnn.benchmark),
                       # world size as the process group itself reports it (a SCALE record can be checked against it)
                       "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
                       # device time of the step's host-launched tail (exchange + clip/AdamW + re-pack), median of 5 steps
                       "tail_ms_per_step": head["tail_ms"],
                       "backend": (dist.get_backend() if world > 1 else None),
                       "noise": "pre-drawn on the host" if args.host_noise else
                                "drawn on the device inside the step (Philox4x32-10 prologue kernel, rank-invariant)"},
        }
        if prof:
            lora = {k: v for k, v in prof.items() if _is_lora_kind(k)}
            name, d = max(lora.items(), key=lambda kv: kv[1]["ms"])
            secs = d["ms"] / 1e3
            ai = d["flops"] / d["bytes"]
            ridge = MFMA_PEAK_TFLOPS[args.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
            if ai < ridge:
                roof = {"bound": "hbm", "achieved": d["bytes"] / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            else:
                roof = {"bound": "mfma", "achieved": d["flops"] / secs / 1e12, "peak": MFMA_PEAK_TFLOPS[args.dtype],
                        "unit": "TFLOP/s"}
            roof["frac"] = roof["achieved"] / roof["peak"]
            if roof["bound"] == "hbm":
                roof["frac_of_achievable_hbm"] = roof["achieved"] / HBM_ACHIEVABLE_GBS  # the /6.29 TB/s view of SURVEY §8(d)
            roof["traffic"] = measured_traffic(name, args.dtype)
            roof.update({"kernel": name, "launches": d["launches"], "avg_us": 1e3 * d["ms"] / d["launches"],
                         "scope": "SURVEY §8(d) layers only: forward / dX launches of LoraInjectedLinear layers (grouped q/k/v "
                                  "counted once per group, gated `proj` forward with its [M,F] output); the gated frozen "
                                  "ff.net.2 backward GEMM is a kind of its own (hot_path.kernels) and enters "
                                  "`frac_incl_fused` only",
                         "measured": "dispatch-attached HIP events on the launch stream, second pass over the same K steps "
                                     "(events off in the pass that yields `value`)",
                         "ms_per_step_with_events": 1e3 * elapsed_prof / args.steps,
                         "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                         "algorithmic_flops_per_launch": d["flops"] / d["launches"],
                         "hbm_frac": d["bytes"] / secs / 1e9 / HBM_PEAK_GBS,
                         "mfma_frac": d["flops"] / secs / 1e12 / MFMA_PEAK_TFLOPS[args.dtype]})
            fused = [v for k, v in prof.items() if k.startswith("geglu_linear_bwd")]
            if fused:  # the r02 view: the same class with the 16 gated frozen GEMMs counted in
                b = d["bytes"] + sum(v["bytes"] for v in fused)
                f = d["flops"] + sum(v["flops"] for v in fused)
                t = secs + sum(v["ms"] for v in fused) / 1e3
                roof["frac_incl_fused"] = (b / t / 1e9 / HBM_PEAK_GBS) if ai < ridge else (f / t / 1e12 / MFMA_PEAK_TFLOPS[args.dtype])
            # every fused forward / dX launch of the step, whatever tile class the plan put it in (classes change between
            # rounds; this figure does not): Σ algorithmic bytes ÷ Σ duration
            gemms = [v for k, v in lora.items() if k.startswith("lora_gemm_kernel") and "false" not in k]
            gb, gf, gt = (sum(v[x] for v in gemms) for x in ("bytes", "flops", "ms"))
            roof["all_fused_gemm_launches"] = {"launches_per_step": sum(v["launches"] for v in gemms) / args.steps,
                                               "ms_per_step": gt / args.steps, "hbm_frac": gb / gt / 1e6 / HBM_PEAK_GBS,
                                               "mfma_frac": gf / gt / 1e9 / MFMA_PEAK_TFLOPS[args.dtype]}
            roof["fused_gemm_classes"] = {k: {"launches_per_step": v["launches"] / args.steps, "ms_per_step": v["ms"] / args.steps,
                                              "hbm_frac": v["bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS,
                                              "mfma_frac": v["flops"] / v["ms"] / 1e9 / MFMA_PEAK_TFLOPS[args.dtype]}
                                          for k, v in lora.items() if k.startswith("lora_gemm_kernel")}
            roof["note"] = ("`kernel` is the LoRA kind with the largest total time.  Through round 2 that was the 128-row tile class; "
                            "since round 3 the 30 launches/step of the 320-wide projections run on 64x160 tiles and the split-K launches "
                            "are a kind of their own, so the 64-row class is the largest — compare rounds with `all_fused_gemm_launches` "
                            "(every fused forward / dX launch, whatever its tile) and `fused_gemm_classes`, not across class boundaries")
            # step level: every §8(d) kernel against the time the algorithmic bytes need at the HBM peak
            tot_b = sum(v["bytes"] for v in lora.values()) / args.steps
            tot_ms = sum(v["ms"] for v in lora.values()) / args.steps
            sv = head["survey"]
            sv_b = sv["fwd_bytes"] + sv["bwd_bytes"]
            roof["step_level"] = {
                # the CONTRACT's figure: SURVEY §8(d) per-layer bytes (every operand once per direction; a dX for every layer
                # but the frozen encoder's attn2.to_k/to_v) summed over the layers that ran — 5 331 MB at cfg-2 — against all
                # §8(d) kernel time.  `survey_MB_as_run`: the same sum with a dX only where the step needs one (the first
                # block's q/k/v read a tensor nothing trainable precedes: 32 MB less at cfg-2)
                "algorithmic_MB_per_step_survey": sv_b / 1e6, "survey_fwd_MB": sv["fwd_bytes"] / 1e6,
                "survey_MB_as_run": (head["survey_as_run"]["fwd_bytes"] + head["survey_as_run"]["bwd_bytes"]) / 1e6,
                "survey_bwd_MB": sv["bwd_bytes"] / 1e6, "survey_GF_per_step": (sv["fwd_flops"] + sv["bwd_flops"]) / 1e9,
                "kernel_ms_per_step": tot_ms,
                "hbm_bound_ms": sv_b / (HBM_PEAK_GBS * 1e9) * 1e3,
                "frac": sv_b / (HBM_PEAK_GBS * 1e9) * 1e3 / tot_ms,
                "frac_of_achievable_hbm": sv_b / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3 / tot_ms,
                # what the kernels are CHARGED per launch (lora_prof_*): the survey bytes plus operands a second kernel reads
                # again — the factor-gradient pass re-reads dY and X (the backward formula counts them once) and writes/reads
                # T, U; gated forward launches also write the [M,F] gated output.  Extra traffic, not algorithmic work.
                "charged_MB_per_step": tot_b / 1e6, "extra_traffic_MB_per_step": (tot_b - sv_b) / 1e6,
                "frac_on_charged_bytes": tot_b / (HBM_PEAK_GBS * 1e9) * 1e3 / tot_ms}
            add_launch_floor(roof, head.get("launch_floor") or {}, tot_ms)
            result["roofline"] = roof
            result["hot_path"] = hot_path_summary(prof, args.steps, elapsed_prof)
        if world == 1 and not args.no_extra and args.config == 2:
            extras = []
            for cid in (3, 4, 5):
                try:
                    r = run_workload(args, cid, rank, world, device, dist, profile=not args.no_prof)
                except Exception as exc:  # an extra line must never cost the headline
                    log(f"[cfg-{cid}] failed: {exc!r}")
                    extras.append({"config": cid, "error": repr(exc)})
                    continue
                c = r["cfg"]
                e = {"config": cid, "workload": c["what"].format(**c) + f", {c['latent'] * 8}^2, {args.dtype}",
                     "images_s": c["batch"] * args.steps / r["elapsed"],
                     "rows_s": c["batch"] * r["rows_per_image"] * args.steps / r["elapsed"],
                     "ms_per_step": 1e3 * r["elapsed"] / args.steps, "hipgraph": r["graph_used"],
                     "lora_params": r["lora_params"], "final_loss": r["final_loss"], "overflow": r["overflow"],
                     "launch_mode_trial": r["launch_trial"], "tail_ms_per_step": r["tail_ms"],
                     "survey_MB_per_step": (r["survey"]["fwd_bytes"] + r["survey"]["bwd_bytes"]) / 1e6}
                if c["v_prediction"]:
                    # (say it wherever a v-prediction number is quoted)
                    e["note"] = ("v-prediction target and add_noise use the DDPM scaled-linear schedule restated from the model's hub "
                                 "config / diffusers, which is not part of the reference tree: parity unpinned for those constants")
                if r["prof"]:
                    hp = hot_path_summary(r["prof"], args.steps, r["elapsed_prof"])
                    e["hot_path_ms"] = hp["kernel_ms_per_step"]
                    e["hot_path_ms_incl_f4"] = hp["kernel_ms_per_step_incl_f4"]
                extras.append(e)
            result["extra_configs"] = extras
            try:
                # a fresh process, like a trainer's own: after four workloads in this one the host-bound route measures
                # 10-15 % low (allocator and interpreter state of the previous models)
                import subprocess

                torch.cuda.empty_cache()
                # (the route is host-bound and the host's speed wanders by several percent within seconds on a shared box: at least
                #  30 timed steps — 1.2 s — whatever the headline's K)
                cmd = [sys.executable, os.path.abspath(__file__), "--drop-in", "--steps", str(max(30, args.steps)),
                       "--warmup", str(max(5, args.warmup))]
                out = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, timeout=600)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
                d = json.loads(line)
                result["extra"] = {"drop_in": {"images_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                                               "final_loss": d["final_loss"], "route": d["route"],
                                               "measured_in": "a fresh child process (bench.py --drop-in)"}}
            except Exception as exc:
                log(f"drop-in route in a child process failed ({exc!r}); measuring it in this process")
                try:
                    result["extra"] = {"drop_in": drop_in_route(args, device)}
                except Exception as exc2:
                    result["extra"] = {"drop_in": {"error": repr(exc2)}}
        if world == 1 and not args.no_extra and args.config == 2:
            try:
                result.setdefault("extra", {})["exchange_probe"] = exchange_probe(args, device)
            except Exception as exc:  # (never at the headline's expense)
                log(f"exchange probe failed: {exc!r}")
                result.setdefault("extra", {})["exchange_probe"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            result["cpu_baseline"] = cpu_baseline(CONFIGS[2]["rank"], CONFIGS[2]["latent"], args.cpu_steps)
        print(json.dumps(result), flush=True)
    elif world == 1:
        pass
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
