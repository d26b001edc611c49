"""bf16 at STEP level.  The reference offers `--mixed_precision bf16` (training_scripts/train_lora_dreambooth.py:
409-416: accelerate's autocast, fp32 master weights, NO GradScaler for bf16 — :489-494, 759-763); the kernels are tested in bf16 one by
one (operator matrix, GEGLU, both attention cores, factor gradients) — here the whole step runs in it: `LoraTrainer` on a
`.bfloat16()` UNet (host-launched and recorded), the unchanged-trainer route under `torch.autocast(dtype=torch.bfloat16)`, and one
full-size config-2 step against the fp32 CPU oracle.  bf16 keeps 8 significant bits: the bounds are those of the f16 tests times
the ratio of the two formats' roundings (2^-8 / 2^-11), each at most 2× what the committed kernels measure (values in the comments)."""
import itertools
import json

import pytest
import torch

import diffusion_finetuning_amd as dfa
from diffusion_finetuning_amd import trainer as tr
from diffusion_finetuning_amd.attention import set_use_hip_geglu, set_use_memory_efficient_attention_xformers
from oracle import lora_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _warm(plist, seed, std):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for i, p in enumerate(plist):
            if i % 2 == 0:
                p.copy_(torch.randn(p.shape, generator=g).to(p.device) * std)


@pytest.mark.parametrize("mode", ["eager", "graph", "hooks"])
def test_bf16_trainer_tracks_the_reference_trajectory(golden_trajectory, tiny_unet_factory, relerr, mode):
    """`LoraTrainer` on a bf16 model against the REFERENCE-produced 10-step fp32 trajectory (tests/golden/trajectory.safetensors,
    oracle/make_golden.py): bf16 storage / compute, fp32 master factors, loss scale 1 (bf16 has fp32's exponent range: no scaler,
    train_lora_dreambooth.py:489-494) — host-launched, recorded into a hipGraph, and with the attention cores and the gated GEGLU
    epilogues switched on."""
    t, meta = golden_trajectory
    cfg = json.loads(meta["plain"])
    unet = tiny_unet_factory(seed=cfg["unet_seed"]).bfloat16().to(DEV)
    params, _ = dfa.inject_trainable_lora(unet, r=4)
    plist = list(itertools.chain(*params))
    if mode == "hooks":
        set_use_memory_efficient_attention_xformers(unet, True)
        set_use_hip_geglu(unet, True)
    trainer = tr.LoraTrainer(unet, lr=cfg["lr"], capture_graph=(mode == "graph"))
    assert trainer.loss_scale == 1.0 and trainer.slab.params.dtype == torch.float32 and plist[0].dtype == torch.float32
    _warm(plist, cfg["warm_seed"], cfg["warm_std"])
    losses = []
    for step in range(cfg["steps"]):
        latents, noise, ts, ctx = orc.synthetic_batch(step, cfg["batch"], cfg["latent_hw"], cfg["ctx_len"], cfg["ctx_dim"])
        losses.append(trainer.step(latents.to(DEV), noise.to(DEV), ts.to(DEV), ctx.to(DEV)).item())
    assert not trainer.opt.overflowed() and trainer.opt.applied_steps() == cfg["steps"]
    if mode == "graph":
        assert trainer._graph is not None  # the bf16 step was really recorded and replayed
    lerr = relerr(torch.tensor(losses), t["plain.losses"])
    upd = tr.flat_lora_state(unet).cpu() - t["plain.init"]
    uerr = relerr(upd, t["plain.final"] - t["plain.init"])
    serr = relerr(tr.flat_lora_state(unet), t["plain.final"])
    print(f"bf16 trainer [{mode}] vs the fp32 reference trajectory: losses {lerr:.2e}, update {uerr:.3f}, state {serr:.2e}")
    assert lerr < BOUNDS["traj_loss"], lerr
    assert uerr < BOUNDS["traj_update"], uerr
    assert serr < BOUNDS["traj_state"], serr


def test_bf16_drop_in_route_under_autocast_without_a_scaler(golden_trajectory, tiny_unet_factory, relerr):
    """The unchanged-trainer route as accelerate runs it under `--mixed_precision bf16` (train_lora_dreambooth.py:409-416,489-494):
    fp32 module, `torch.autocast(dtype=torch.bfloat16)` around the forward, `F.mse_loss`-equivalent on `.float()`, plain
    `loss.backward()` (no GradScaler for bf16), `clip_grad_norm_`, torch AdamW over the chained generators — with the reference's
    attention switch on, so the bf16 attention cores, the GEGLU fusion and the grouped projections are what runs."""
    t, meta = golden_trajectory
    cfg = json.loads(meta["plain"])
    unet = tiny_unet_factory(seed=cfg["unet_seed"]).to(DEV)
    params, _ = dfa.inject_trainable_lora(unet, r=4)
    plist = list(itertools.chain(*params))
    _warm(plist, cfg["warm_seed"], cfg["warm_std"])
    set_use_memory_efficient_attention_xformers(unet, True)
    opt = torch.optim.AdamW(plist, lr=cfg["lr"], betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    acp = orc.ddpm_alphas_cumprod()
    seen = []
    for step in range(cfg["steps"]):
        latents, noise, ts, ctx = orc.synthetic_batch(step, cfg["batch"], cfg["latent_hw"], cfg["ctx_len"], cfg["ctx_dim"])
        noisy = orc.add_noise(latents, noise, ts, acp).to(DEV)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = unet(noisy, ts.to(DEV), ctx.to(DEV)).sample
        seen.append(pred.dtype)
        loss = dfa.ddpm_mse_loss(pred.float(), noise.to(DEV).float())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(unet.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
    assert all(p.dtype == torch.float32 for p in plist)  # masters stay fp32 under autocast
    assert all(torch.isfinite(p).all() for p in plist)
    upd = tr.flat_lora_state(unet).cpu() - t["plain.init"]
    uerr = relerr(upd, t["plain.final"] - t["plain.init"])
    serr = relerr(tr.flat_lora_state(unet), t["plain.final"])
    print(f"bf16 drop-in route (autocast, no scaler) vs the fp32 reference trajectory: update {uerr:.3f}, state {serr:.2e}; "
          f"prediction dtype {seen[0]}")
    assert uerr < BOUNDS["traj_update"], uerr
    assert serr < BOUNDS["traj_state"], serr


def test_full_size_cfg2_bf16_step_vs_cpu_oracle(relerr):
    """BASELINE config 2 at full size in bf16 — SD1.5-shaped UNet, batch 4, 64×64 latents, every fused path on (grouped
    projections, both attention cores, gated GEGLU epilogues, fused loss, clip + AdamW) — ONE step against the fp32 CPU oracle on
    the same weights and inputs: loss, direction of the gradient slab (whole and worst layer), signs of the update weighted by
    |g| (Adam's first step is ≈ lr·sign(g): the state itself would pass un-updated)."""
    import bench
    from tests.test_gpu_groups import _sd15, weighted_sign_agreement

    torch.set_num_threads(bench.usable_cpus())
    ref = _sd15("cpu", torch.float32)
    ref_params, _ = orc.inject(ref, r=4)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for i, p in enumerate(ref_params):
            if i % 2 == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.01)
    init_state = orc.flat_params(ref_params).clone()
    state = {k: v.clone() for k, v in ref.state_dict().items() if "lora_" not in k}
    ref_loss = orc.train_steps(ref, ref_params, 1, 4, 64, 77, 768, lr=1e-4)[0]
    ref_grad = torch.cat([p.grad.reshape(-1) for p in ref_params])
    want = orc.flat_params(ref_params)
    del ref

    unet = _sd15("cpu", torch.float32)
    unet.load_state_dict({k.replace(".linear.", "."): v for k, v in state.items()})
    unet = unet.bfloat16().to(DEV)
    params, _ = dfa.inject_trainable_lora(unet, r=4)
    plist = list(itertools.chain(*params))
    with torch.no_grad():
        for p, rp in zip(plist, torch.split(init_state, [q.numel() for q in plist])):
            p.copy_(rp.view(p.shape).to(DEV))
    set_use_memory_efficient_attention_xformers(unet, True)
    set_use_hip_geglu(unet, True)
    trainer = tr.LoraTrainer(unet, lr=1e-4)
    assert trainer.loss_scale == 1.0 and len(trainer.slab.qkv_groups) == 16 and trainer.slab.ctx_groups[0].G == 32
    lat, noise, ts, ctx = orc.synthetic_batch(0, 4, 64, 77, 768)
    loss = trainer.step(lat.to(DEV), noise.to(DEV), ts.to(DEV), ctx.to(DEV)).item()
    assert not trainer.opt.overflowed() and trainer.opt.applied_steps() == 1
    grad = trainer.slab.grads[: trainer.slab.numel].cpu()
    got = tr.flat_lora_state(unet).cpu()
    lerr = abs(loss - ref_loss) / abs(ref_loss)
    gn, rn = grad / grad.norm(), ref_grad / ref_grad.norm()
    derr = relerr(gn, rn)
    worst = max(relerr(gn[o:o + n], rn[o:o + n]) for o, n in trainer.slab.offsets)
    agree = (((got - init_state) * (want - init_state)) > 0).float().mean().item()
    wsign = weighted_sign_agreement(got - init_state, want - init_state, ref_grad)
    print(f"cfg-2 bf16, one step vs the fp32 oracle: loss err {lerr:.2e}, gradient direction err {derr:.2e} (worst layer "
          f"{worst:.2e}), update signs agree on {agree:.4f} of the elements / {wsign:.4f} of the gradient mass")
    assert lerr < BOUNDS["full_loss"], lerr
    assert derr < BOUNDS["full_direction"] and worst < BOUNDS["full_worst_layer"], (derr, worst)
    assert agree > BOUNDS["full_sign_elements"] and wsign > BOUNDS["full_sign_mass"], (agree, wsign)
    assert float((got - init_state).abs().max()) > 0.5e-4  # the step was applied (|Δ| ≈ lr = 1e-4)


# at most 2× what the committed kernels measure (round 6, one box: trajectory losses 3.7e-4 / 4.9e-4 with the hooks, update
# 0.074 (drop-in route 0.046), state 2.1e-3; full size: loss 4.9e-4, direction 8.9e-3, worst layer 8.9e-2, update signs 0.873 of
# the elements / 0.987 of the gradient mass — the f16 step measures 1.3e-3 / 2.2e-2 / 0.99 / 0.998: three bits fewer)
BOUNDS = {"traj_loss": 1e-3, "traj_update": 0.15, "traj_state": 4.5e-3, "full_loss": 1e-3, "full_direction": 1.8e-2,
          "full_worst_layer": 0.18, "full_sign_elements": 0.82, "full_sign_mass": 0.975}
