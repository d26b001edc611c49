"""Dev tool: where the HOST time of the unchanged-trainer route goes (bench.py::drop_in_route's step).
Wall-clock sections of the step without a sync in between (forward / backward / optimizer enqueue), then µs per call of the
library's entry points on that route — LoraInjectedLinear.forward, the attention hook, the feed-forward hook, and the backward of
each autograd Function — measured with perf_counter wrappers (no cProfile: its per-call overhead distorts 10-µs functions).
usage: python tools/dropin_sections.py [steps]"""
import collections
import itertools
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
import diffusion_finetuning_amd as dfa  # noqa: E402
from diffusion_finetuning_amd import attention, groups, ops  # noqa: E402
from diffusion_finetuning_amd.attention import set_use_memory_efficient_attention_xformers  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CPROFILE = len(sys.argv) > 2 and sys.argv[2] == "cprofile"  # also: cProfile of 4 steps, top entries by own time
args = types.SimpleNamespace(warmup=3, steps=steps, no_conv_autotune=False)
bench.conv_autotune(args)
device = torch.device("cuda", 0)
cfg = bench.CONFIGS[2]
from harness.unet import UNet2DConditionModel, sd15_config  # noqa: E402

torch.manual_seed(0)
unet = UNet2DConditionModel(sd15_config()).to(device)
unet.requires_grad_(False)
params, _ = dfa.inject_trainable_lora(unet, r=cfg["rank"])
set_use_memory_efficient_attention_xformers(unet, True)
plist = list(itertools.chain(*params))
opt = torch.optim.AdamW(plist, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
from diffusion_finetuning_amd.trainer import ddpm_tables  # noqa: E402

sa, sb = ddpm_tables(device=device)
data = bench.synthetic_steps(args.warmup + 2 * steps, cfg["batch"], cfg["latent"], 0, 1, device)

acc = collections.defaultdict(lambda: [0, 0.0, []])
# the autograd nodes of the route (names differ between the trees this tool compares: take what the tree has)
FUNCTIONS = [getattr(m, n) for m, n in [(ops, "_LoraLinearFn"), (ops, "_LoraGegluFn"), (ops, "_LoraProjGatedFn"), (ops, "_GatedLinearFn"),
                                         (ops, "_FeedForwardFn"), (groups, "_QKVProjFn"), (groups, "_FlashQKVFn"), (groups, "_QKVAttnFn"),
                                         (groups, "_CtxProjFn"), (groups, "_CtxAttnKVFn")] if hasattr(m, n)]


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            e = acc[name]
            e[0] += 1
            e[1] += dt
            e[2].append(dt)
    return wrapper


sec = collections.defaultdict(float)


def step(i, clock=False):
    lat, _, _, ctx = data[i]
    t0 = time.perf_counter()
    noise = torch.randn_like(lat)
    t = torch.randint(0, 1000, (lat.shape[0],), device=device)
    noisy = sa[t].view(-1, 1, 1, 1) * lat + sb[t].view(-1, 1, 1, 1) * noise
    with torch.autocast("cuda", dtype=torch.float16):
        pred = unet(noisy, t, ctx).sample
    loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    t1 = time.perf_counter()
    scaler.scale(loss).backward()
    t2 = time.perf_counter()
    scaler.unscale_(opt)
    torch.nn.utils.clip_grad_norm_(plist, 1.0)
    scaler.step(opt)
    scaler.update()
    opt.zero_grad()
    t3 = time.perf_counter()
    if clock:
        sec["forward + loss"] += t1 - t0
        sec["backward"] += t2 - t1
        sec["unscale + clip + AdamW + zero_grad"] += t3 - t2
    return loss


for i in range(args.warmup):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
per_step = []
for i in range(args.warmup, args.warmup + steps):
    ts = time.perf_counter()
    step(i, clock=True)
    per_step.append(1e3 * (time.perf_counter() - ts))
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"unchanged-trainer step: enqueue {1e3 * t_enq / steps:.2f} ms, drained {1e3 * t_all / steps:.2f} ms per step "
      f"({cfg['batch'] * steps / t_all:.1f} images/s)")
print("  enqueue ms of every step: " + " ".join(f"{v:.1f}" for v in per_step) + f"   (fastest {min(per_step):.2f}, median {sorted(per_step)[len(per_step) // 2]:.2f})")
for k, v in sec.items():
    print(f"  host time in {k:38s} {1e3 * v / steps:7.2f} ms per step")

# ---- per-call host time of the library's entry points (wrappers installed AFTER the plain timing above) -------------------
ops.lora_linear = timed("ops.lora_linear (LoraInjectedLinear.forward body)", ops.lora_linear)
import diffusion_finetuning_amd.core as core  # noqa: E402

core.lora_linear = ops.lora_linear  # (core.py binds the name at import)
attention._hip_forward_orig = attention._hip_forward
for m in unet.modules():
    f = m.__dict__.get("forward")
    if f is not None and getattr(f, "func", None) is attention._hip_forward:
        m.__dict__["forward"] = timed("attention hook (to_q/k/v + core + to_out)", f)
    elif f is not None and getattr(f, "func", None) is getattr(attention, "_hip_feed_forward", None):
        m.__dict__["forward"] = timed("feed-forward hook (proj + gate + net.2)", f)
for cls in FUNCTIONS:
    cls.backward = staticmethod(timed(f"{cls.__name__}.backward", cls.backward))
if len(sys.argv) > 2 and sys.argv[2] == "fine":  # finer grain: forward bodies, the group entry points, the ctypes wrappers
    from diffusion_finetuning_amd import _native as nat

    for cls in FUNCTIONS:
        cls.forward = staticmethod(timed(f"{cls.__name__}.forward (body only, inside apply)", cls.forward))
    for fn in ("qkv_self_attention", "ctx_cross_attention"):
        wrapped = timed(f"groups.{fn} (apply calls included)", getattr(groups, fn))
        setattr(groups, fn, wrapped)
        setattr(attention, fn, wrapped)
    for fn in ("lora_linear_fwd", "lora_linear_geglu_fwd", "lora_linear_bwd_input", "lora_gemm_packed", "attn_flash_fwd_qkv",
               "attn_flash_bwd_qkv", "attn_ctx_fwd_kv", "attn_ctx_bwd_kv", "geglu_linear_bwd", "lora_gemm_parts"):
        if hasattr(nat, fn):
            setattr(nat, fn, timed(f"_native.{fn} (allocations + one ctypes call)", getattr(nat, fn)))
for name in ("usable",):
    groups.QKVGroup.usable = timed("QKVGroup.usable", groups.QKVGroup.usable)
    groups.CtxKVGroup.usable = timed("CtxKVGroup.usable", groups.CtxKVGroup.usable)
ops.PackRegistry._repack = timed("PackRegistry._repack (once per optimizer step: every layer's packed factors)", ops.PackRegistry._repack)
ops._AutoSink.flush = timed("_AutoSink.flush (end of backward: batched factor gradients + .grad hand-over)", ops._AutoSink.flush)
for i in range(args.warmup + steps, args.warmup + 2 * steps):
    step(i)
torch.cuda.synchronize()
print("per call (perf_counter wrappers; inner wrappers are included in the outer ones; the MEDIAN call is what compares between runs — "
      "the host's speed drifts by ±10 % within a minute on a shared box):")
for k, (n, t, each) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    each.sort()
    print(f"  {k:82s} {n / steps:6.1f} calls/step  {1e6 * t / n:8.1f} us each (median {1e6 * each[len(each) // 2]:6.1f})  "
          f"{1e3 * t / steps:7.2f} ms per step")

if CPROFILE:
    import cProfile
    import pstats

    prof = cProfile.Profile()
    prof.enable()
    for i in range(args.warmup, args.warmup + 4):
        step(i)
    prof.disable()
    torch.cuda.synchronize()
    pstats.Stats(prof).sort_stats("tottime").print_stats(60)
