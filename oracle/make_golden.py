"""Generates tests/golden/* by running the REFERENCE's own code (build container only).

    python oracle/make_golden.py            # needs /root/reference; never runs on the GPU box

It imports the reference's `lora_diffusion/lora.py` standalone (its package __init__ pulls cv2/diffusers, which
are absent), runs it on seeded inputs on the CPU exactly as the reference trainers use it (torch AdamW,
clip_grad_norm_, F.mse_loss), and stores INPUTS AND EXPECTED OUTPUTS as small fixtures.  No reference source
text is written anywhere.  Big operands are rounded to fp16-representable values before use so the same
fixture is an exact input for the f16 and f32 kernels alike.
"""
import importlib
import itertools
import json
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F
from safetensors import safe_open
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference_lora():
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("lora_diffusion")
    pkg.__path__ = [os.path.join(REF, "lora_diffusion")]
    sys.modules["lora_diffusion"] = pkg
    return importlib.import_module("lora_diffusion.lora")


def h16(t):
    """Round to fp16-representable values, keep fp32 storage."""
    return t.half().float()


OPERATOR_CASES = [
    # M, K, N, r, bias, scale
    (64, 32, 32, 1, False, 1.0),
    (64, 32, 64, 4, True, 0.7),
    (77, 96, 64, 8, True, 1.0),
    (64, 64, 128, 16, False, 0.7),
    (50, 36, 20, 3, True, 1.0),      # 16-bit path unaligned → generic kernels
    (33, 30, 18, 2, False, 0.7),     # unaligned for every dtype
    (64, 320, 320, 4, True, 1.0),    # SD1.5 attn1 shape
    (77, 768, 320, 4, False, 0.7),   # SD1.5 attn2 to_k/to_v shape
]


def gen_operator(ref):
    tensors, meta = {}, {}
    g = torch.Generator().manual_seed(1234)
    for ci, (M, K, N, r, bias, scale) in enumerate(OPERATOR_CASES):
        layer = ref.LoraInjectedLinear(K, N, bias, r)
        with torch.no_grad():
            layer.linear.weight.copy_(h16((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5))
            if bias:
                layer.linear.bias.copy_(h16(torch.randn(N, generator=g) * 0.1))
            layer.lora_down.weight.copy_(h16(torch.randn(r, K, generator=g) / r))
            layer.lora_up.weight.copy_(h16(torch.randn(N, r, generator=g) * 0.05))
        layer.scale = scale
        layer.linear.requires_grad_(False)
        x = h16(torch.randn(1, M, K, generator=g))
        x.requires_grad_(True)
        dy = h16(torch.randn(*x.shape[:-1], N, generator=g))
        y = layer(x)
        y.backward(dy)
        p = f"c{ci}"
        tensors[f"{p}.x"] = x.detach().half()
        tensors[f"{p}.w"] = layer.linear.weight.detach().half()
        if bias:
            tensors[f"{p}.b"] = layer.linear.bias.detach().half()
        tensors[f"{p}.down"] = layer.lora_down.weight.detach().clone()
        tensors[f"{p}.up"] = layer.lora_up.weight.detach().clone()
        tensors[f"{p}.dy"] = dy.half()
        tensors[f"{p}.y"] = y.detach().clone()
        tensors[f"{p}.dx"] = x.grad.clone()
        tensors[f"{p}.g_down"] = layer.lora_down.weight.grad.clone()
        tensors[f"{p}.g_up"] = layer.lora_up.weight.grad.clone()
        meta[p] = json.dumps({"M": M, "K": K, "N": N, "r": r, "bias": bias, "scale": scale})
    # a1: init statistics + rank error text
    torch.manual_seed(7)
    init = ref.LoraInjectedLinear(320, 320, False, 4)
    meta["init"] = json.dumps({"down_std": float(init.lora_down.weight.std()), "up_absmax": float(init.lora_up.weight.abs().max())})
    try:
        ref.LoraInjectedLinear(8, 16, False, 9)
    except ValueError as e:
        meta["rank_error"] = str(e)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, "operator.safetensors"), meta)


def matrix_digest(y, dx, g_down, g_up, K, N, seed):
    """What operator_matrix.safetensors keeps of one case's outputs (float64 in, small tensors out): the first and last
    rows of Y and dX in full, and the projections of all four outputs onto seeded probe matrices (oracle/synthetic.probes) —
    the wide outputs themselves would be 24 MB over the 96 cases."""
    from oracle import synthetic as syn

    pn, pk = syn.probes(N, seed), syn.probes(K, seed + 1)
    return {"y.rows": y[[0, -1]].float(), "dx.rows": dx[[0, -1]].float(), "y.proj": y.double() @ pn, "dx.proj": dx.double() @ pk,
            "g_down.proj": g_down.double() @ pk, "g_up.proj": pn.t() @ g_up.double()}


def gen_operator_matrix(ref):
    """SURVEY §8(c)'s operator matrix: the REFERENCE's LoraInjectedLinear (lora.py:32-50, run in float64) and autograd on every
    SD1.5 LoRA layer kind — the 9 (K, N) pairs, square ones with and without bias as diffusers has them — × r ∈ {1, 4, 8, 16}
    × scale ∈ {1, 0.7}; inputs are oracle/synthetic.py's integer-hash operands (fp16-exact, not stored)."""
    from oracle import synthetic as syn

    tensors, meta = {}, {}
    for tag, K, N, bias, M, r, scale, seed in syn.matrix_cases():
        x, w, b, dy, down, up = syn.matrix_inputs(K, N, bias, M, r, seed)
        layer = ref.LoraInjectedLinear(K, N, bias, r).double()
        with torch.no_grad():
            layer.linear.weight.copy_(w)
            if bias:
                layer.linear.bias.copy_(b)
            layer.lora_down.weight.copy_(down)
            layer.lora_up.weight.copy_(up)
        layer.scale = scale
        layer.linear.requires_grad_(False)
        xin = x.double().requires_grad_(True)
        y = layer(xin)
        y.backward(dy.double())
        d = matrix_digest(y.detach(), xin.grad, layer.lora_down.weight.grad, layer.lora_up.weight.grad, K, N, seed)
        tensors.update({f"{tag}.{k}": v for k, v in d.items()})
        meta[tag] = json.dumps({"M": M, "K": K, "N": N, "r": r, "bias": bias, "scale": scale, "seed": seed})
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, "operator_matrix.safetensors"), meta)


def gen_losses():
    g = torch.Generator().manual_seed(99)
    tensors = {}
    # a8 plain + prior (train_lora_dreambooth.py:855-875)
    pred = h16(torch.randn(4, 4, 8, 8, generator=g)).requires_grad_(True)
    target = h16(torch.randn(4, 4, 8, 8, generator=g))
    loss = F.mse_loss(pred.float(), target.float(), reduction="mean")
    loss.backward()
    tensors.update({"plain.pred": pred.detach().half(), "plain.target": target.half(), "plain.loss": loss.detach().reshape(1),
                    "plain.dpred": pred.grad.clone()})
    pred2 = h16(torch.randn(6, 4, 8, 8, generator=g)).requires_grad_(True)
    target2 = h16(torch.randn(6, 4, 8, 8, generator=g))
    mp, mpp = torch.chunk(pred2, 2, dim=0)
    tg, tgp = torch.chunk(target2, 2, dim=0)
    l2 = F.mse_loss(mp.float(), tg.float(), reduction="none").mean([1, 2, 3]).mean() + 0.8 * F.mse_loss(
        mpp.float(), tgp.float(), reduction="mean")
    l2.backward()
    tensors.update({"prior.pred": pred2.detach().half(), "prior.target": target2.half(), "prior.loss": l2.detach().reshape(1),
                    "prior.dpred": pred2.grad.clone()})
    # a9 masked (cli_lora_pti.py:222-247)
    pred3 = h16(torch.randn(2, 4, 8, 8, generator=g)).requires_grad_(True)
    target3 = h16(torch.randn(2, 4, 8, 8, generator=g))
    raw = (torch.rand(2, 64, 64, generator=g) > 0.6).float()
    mask = raw.reshape(2, 1, 64, 64)
    mask = F.interpolate(mask.float(), size=pred3.shape[-2:], mode="nearest") + 0.05
    mask = mask / mask.mean()
    l3 = F.mse_loss((pred3 * mask).float(), (target3 * mask).float(), reduction="mean")
    l3.backward()
    tensors.update({"masked.pred": pred3.detach().half(), "masked.target": target3.half(), "masked.raw_mask": raw,
                    "masked.mask": mask.clone(), "masked.loss": l3.detach().reshape(1), "masked.dpred": pred3.grad.clone()})
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, "losses.safetensors"),
              {"prior_loss_weight": "0.8"})


def build_tiny_unet(seed=0):
    from harness.unet import UNet2DConditionModel, tiny_config

    torch.manual_seed(seed)
    unet = UNet2DConditionModel(tiny_config(32, 32, 2))
    unet.requires_grad_(False)
    return unet


def gen_finder_and_formats(ref, tmpdir):
    from harness.unet import UNet2DConditionModel, sd15_config

    out = {}
    # (1) the shipped example pins the 144-entry SD1.5 index → (K, N) table and the key/metadata format
    f = safe_open(os.path.join(REF, "example_loras", "lora_disney.safetensors"), "pt")
    md = f.metadata()
    table = []
    for i in range(144):
        table.append([f.get_tensor(f"unet:{i}:down").shape[1], f.get_tensor(f"unet:{i}:up").shape[0]])
    out["lora_disney"] = {
        "unet_index_KN": table,
        "metadata_non_rank": {k: v for k, v in md.items() if not k.endswith(":rank")},
        "n_keys": len(list(f.keys())),
        "rank_values": sorted(set(v for k, v in md.items() if k.endswith(":rank"))),
        "text_encoder_shapes": [[list(f.get_tensor(f"text_encoder:{i}:up").shape), list(f.get_tensor(f"text_encoder:{i}:down").shape)]
                                for i in range(3)],
        "dtype": str(f.get_tensor("unet:0:up").dtype),
    }
    pt = torch.load(os.path.join(REF, "example_loras", "analog_svd_distill.text_encoder.pt"), map_location="cpu", weights_only=True)
    out["analog_pt"] = {"len": len(pt), "first_shapes": [list(t.shape) for t in pt[:4]], "dtype": str(pt[0].dtype),
                        "type": type(pt).__name__}
    # (2) reference finder on the build-owned SD1.5-shaped UNet (meta device) and the tiny UNet
    with torch.device("meta"):
        big = UNet2DConditionModel(sd15_config())
    names = {id(m): n for n, m in big.named_modules()}
    out["sd15_order"] = [[names[id(mod)], mod.in_features, mod.out_features, mod.bias is not None]
                         for _, _, mod in ref._find_modules(big, ref.DEFAULT_TARGET_REPLACE)]
    tiny = build_tiny_unet()
    names = {id(m): n for n, m in tiny.named_modules()}
    out["tiny_order"] = [[names[id(mod)], mod.in_features, mod.out_features, mod.bias is not None]
                         for _, _, mod in ref._find_modules(tiny, ref.DEFAULT_TARGET_REPLACE)]
    # (3) reference injection: returned names, state_dict keys, sharing, generator count
    params, inj_names = ref.inject_trainable_lora(tiny, r=4)
    out["tiny_inject"] = {"names": inj_names, "n_generators": len(params),
                          "state_dict_keys": [k for k in tiny.state_dict().keys() if "attn1.to_q" in k][:3],
                          "n_lora_params": sum(p.numel() for p in itertools.chain(*params))}
    # (4) reference writers on that model: safetensors keys+metadata, .pt structure
    st_path = os.path.join(tmpdir, "tiny.safetensors")
    ref.save_safeloras({"unet": (tiny, ref.DEFAULT_TARGET_REPLACE)}, st_path)
    g = safe_open(st_path, "pt")
    out["tiny_safetensors"] = {"keys": sorted(g.keys()), "metadata": dict(g.metadata())}
    pt_path = os.path.join(tmpdir, "tiny.pt")
    ref.save_lora_weight(tiny, pt_path)
    lst = torch.load(pt_path, weights_only=True)
    out["tiny_pt"] = {"len": len(lst), "dtype": str(lst[0].dtype), "shapes": [list(t.shape) for t in lst[:4]]}
    # (5) CLIP text-encoder order on a random-init tiny CLIP (transformers)
    try:
        from transformers import CLIPTextConfig, CLIPTextModel

        clip = CLIPTextModel(CLIPTextConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                            num_attention_heads=2, vocab_size=100, max_position_embeddings=16))
        names = {id(m): n for n, m in clip.named_modules()}
        out["clip_order"] = [names[id(mod)] for _, _, mod in ref._find_modules(clip, ref.TEXT_ENCODER_DEFAULT_TARGET_REPLACE)]
    except Exception as e:  # pragma: no cover
        out["clip_order"] = None
        out["clip_order_error"] = repr(e)
    with open(os.path.join(OUT, "structure.json"), "w") as fh:
        json.dump(out, fh, indent=1)


def gen_merge(ref):
    tensors = {}
    g = torch.Generator().manual_seed(5)

    class Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.blk = nn.ModuleDict()

    class CrossAttention(nn.Module):
        def __init__(self):
            super().__init__()
            self.to_q = nn.Linear(48, 64, bias=False)
            self.to_out = nn.ModuleList([nn.Linear(64, 48)])

    w_q = h16((torch.rand(64, 48, generator=g) * 2 - 1) / 7)
    w_o = h16((torch.rand(48, 64, generator=g) * 2 - 1) / 8)
    ups = [h16(torch.randn(64, 4, generator=g) * 0.1), h16(torch.randn(48, 4, generator=g) * 0.1)]
    downs = [h16(torch.randn(4, 48, generator=g) / 4), h16(torch.randn(4, 64, generator=g) / 4)]
    tensors.update({"w_q": w_q, "w_o": w_o, "up0": ups[0], "down0": downs[0], "up1": ups[1], "down1": downs[1]})
    for alpha in (0.5, 1.0, 1.2):
        for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
            m = CrossAttention()
            with torch.no_grad():
                m.to_q.weight.copy_(w_q)
                m.to_out[0].weight.copy_(w_o)
            m = m.to(dt)
            ref.weight_apply_lora(m, [ups[0].clone(), downs[0].clone(), ups[1].clone(), downs[1].clone()], alpha=alpha)
            tensors[f"merged_q.{tag}.a{alpha}"] = m.to_q.weight.detach().float().clone()
            tensors[f"merged_o.{tag}.a{alpha}"] = m.to_out[0].weight.detach().float().clone()
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, "merge.safetensors"))


def gen_trajectory(ref):
    """Row H: 10 steps of the reference loop (train_lora_dreambooth.py:811-888) on the tiny UNet with the
    REFERENCE's injection, torch.optim.AdamW and clip_grad_norm_; synthetic latents as in oracle.synthetic_batch."""
    from oracle import lora_oracle as orc

    tensors, meta = {}, {}
    for tag, with_prior, batch in (("plain", False, 2), ("prior", True, 4)):
        unet = build_tiny_unet(seed=3)
        params, _ = ref.inject_trainable_lora(unet, r=4)
        plist = list(itertools.chain(*params))
        # warm-start `up` so the first steps are not on the all-zero branch
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for i, p in enumerate(plist):
                if i % 2 == 0:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        tensors[f"{tag}.init"] = orc.flat_params(plist).clone()
        opt = torch.optim.AdamW(plist, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
        acp = orc.ddpm_alphas_cumprod()
        losses = []
        for step in range(10):
            latents, noise, t, ctx = orc.synthetic_batch(step, batch, 8, 6, 32)
            noisy = orc.add_noise(latents, noise, t, acp)
            pred = unet(noisy, t, ctx).sample
            if with_prior:
                mp, mpp = torch.chunk(pred, 2, dim=0)
                tg, tgp = torch.chunk(noise, 2, dim=0)
                loss = F.mse_loss(mp.float(), tg.float(), reduction="none").mean([1, 2, 3]).mean()
                loss = loss + 1.0 * F.mse_loss(mpp.float(), tgp.float(), reduction="mean")
            else:
                loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
            loss.backward()
            torch.nn.utils.clip_grad_norm_(unet.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            losses.append(loss.item())
        tensors[f"{tag}.final"] = orc.flat_params(plist).clone()
        tensors[f"{tag}.losses"] = torch.tensor(losses)
        meta[tag] = json.dumps({"batch": batch, "latent_hw": 8, "ctx_len": 6, "ctx_dim": 32, "lr": 1e-3, "steps": 10,
                                "with_prior": with_prior, "unet_seed": 3, "warm_seed": 11, "warm_std": 0.02})
    save_file(tensors, os.path.join(OUT, "trajectory.safetensors"), meta)


def gen_pti_trajectory(ref, linear_schedule=False):
    """(linear_schedule: the run as perform_tuning really schedules it by default — `get_scheduler("linear", optimizer,
    num_warmup_steps=0, num_training_steps=max_train_steps_tuning)`, cli_lora_pti.py:534-535,746-751, stepped BEFORE every
    batch, :434.  diffusers is absent here; its "linear" is a torch LambdaLR whose λ is restated below from the published
    definition (parity unpinned for that formula) — torch's LambdaLR, AdamW and the reference's LoRA modules do the rest.
    Written to pti_trajectory_linear.safetensors: results only, the inputs are pti_trajectory.safetensors'.)
    BASELINE config 5's second half — the PTI tuning phase with continue_inversion (cli_lora_pti.py:693-753): the REFERENCE's
    inject_trainable_lora on the tiny UNet, the token table of a tiny transformers CLIPTextModel as the second AdamW group
    (:706-722,738), loss_step's arithmetic (:170-248: draw below int(1000·0.8), text encoder inside the step, v-prediction
    target, mse), clip_grad_norm_ over chain(unet.parameters(), text_encoder.parameters()) (:448-450).  cli_lora_pti.py itself
    needs diffusers / fire / wandb and cannot be imported here: the loop below is its step spelled with the same torch calls
    around the reference's own LoRA modules."""
    from transformers import CLIPTextConfig, CLIPTextModel

    from oracle import lora_oracle as orc

    vocab, ctx_len, steps, batch = 60, 8, 8, 2
    torch.manual_seed(21)
    te = CLIPTextModel(CLIPTextConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                      vocab_size=vocab, max_position_embeddings=ctx_len, bos_token_id=1, eos_token_id=2,
                                      pad_token_id=0))
    unet = build_tiny_unet(seed=3)
    params, _ = ref.inject_trainable_lora(unet, r=4)                     # :693
    plist = list(itertools.chain(*params))
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for i, p in enumerate(plist):
            if i % 2 == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    te.requires_grad_(False)                                             # :704
    te.requires_grad_(True)                                              # :717 (continue_inversion)
    table = te.get_input_embeddings().weight
    for name, p in te.named_parameters():                                # :718-724: everything but the token table frozen again
        if p is not table:
            p.requires_grad = False
    tensors = {"lora.init": orc.flat_params(plist).clone(), "table.init": table.detach().clone()}
    tensors.update({f"te.{n}": p.detach().clone() for n, p in te.named_parameters() if p is not table})
    opt = torch.optim.AdamW([{"params": plist, "lr": 1e-3}, {"params": te.get_input_embeddings().parameters(), "lr": 5e-3}],
                            weight_decay=1e-3)                           # :726-738 (weight_decay_lora)
    sched = None
    if linear_schedule:                                                  # :746-751
        warm, total = 0, steps
        sched = torch.optim.lr_scheduler.LambdaLR(
            opt, lambda e: float(e) / float(max(1, warm)) if e < warm else max(0.0, float(total - e) / float(max(1, total - warm))))
    acp = orc.ddpm_alphas_cumprod()
    losses, all_ids, lrs = [], [], []
    for step in range(steps):
        if sched is not None:
            sched.step()                                                 # :434
            lrs.append(sched.get_last_lr())
        latents, noise, t, _ = orc.synthetic_batch(step, batch, 8, ctx_len, 32, t_max=int(1000 * 0.8))   # :190-195, :444
        ids = orc.synthetic_token_ids(step, batch, ctx_len, vocab)
        all_ids.append(ids)
        opt.zero_grad()                                                  # :436
        noisy = orc.add_noise(latents, noise, t, acp)
        ehs = te(ids)[0]                                                 # :199-206
        pred = unet(noisy, t, ehs).sample
        target = orc.get_velocity(latents, noise, t, acp)                # :217-218
        loss = F.mse_loss(pred.float(), target.float(), reduction="mean")  # :247
        loss.backward()                                                  # :447
        torch.nn.utils.clip_grad_norm_(itertools.chain(unet.parameters(), te.parameters()), 1.0)  # :448-450
        opt.step()                                                       # :451
        losses.append(loss.item())
    tensors["lora.final"] = orc.flat_params(plist).clone()
    tensors["table.final"] = table.detach().clone()
    tensors["losses"] = torch.tensor(losses)
    if linear_schedule:
        out = {"lora.final": tensors["lora.final"], "table.final": tensors["table.final"], "losses": tensors["losses"],
               "lrs": torch.tensor(lrs, dtype=torch.float64)}
        save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(OUT, "pti_trajectory_linear.safetensors"),
                  {"schedule": json.dumps({"name": "linear", "num_warmup_steps": 0, "num_training_steps": steps,
                                           "stepped": "before the optimizer (cli_lora_pti.py:434)"})})
        return
    tensors["ids"] = torch.stack(all_ids)
    meta = {"cfg": json.dumps({"vocab": vocab, "ctx_len": ctx_len, "steps": steps, "batch": batch, "latent_hw": 8,
                               "hidden": 32, "intermediate": 64, "layers": 2, "heads": 2, "lr_unet": 1e-3, "lr_embed": 5e-3,
                               "weight_decay": 1e-3, "unet_seed": 3, "warm_seed": 11, "warm_std": 0.02, "t_multiplier": 0.8,
                               "v_prediction": True})}
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, "pti_trajectory.safetensors"), meta)


def main():
    import tempfile

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = import_reference_lora()
    gen_operator(ref)
    gen_operator_matrix(ref)
    gen_losses()
    gen_merge(ref)
    with tempfile.TemporaryDirectory() as d:
        gen_finder_and_formats(ref, d)
    gen_trajectory(ref)
    gen_pti_trajectory(ref)
    gen_pti_trajectory(ref, linear_schedule=True)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
