"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU restatement of the reference algorithm for the LoRA hot path of
levayz/diffusion_finetuning.  It exists to CHECK the HIP path; nothing in the product package
(`diffusion_finetuning_amd/`) imports it.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may use it.

Pinning: `oracle/make_golden.py` (run in the build container, where /root/reference exists) imports the
reference's own `lora_diffusion/lora.py` and torch's AdamW / clip_grad_norm_ / mse_loss exactly as the
reference trainers call them, and writes input+output vectors to `tests/golden/`.
`tests/test_oracle_golden.py` asserts this restatement reproduces every one of them, and the reference's
shipped `example_loras/*` pin the enumeration order and file formats.  The DDPM schedule (`add_noise`,
`get_velocity`) lives in the un-vendored `diffusers` package: that part is restated from the published DDPM
definitions and is "parity unpinned" (no fixture in the reference covers it).

Every function cites the reference lines it follows (paths relative to the reference repo).
"""
import itertools
import math
from typing import List, Optional, Sequence, Set, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

UNET_TARGETS = {"CrossAttention", "Attention", "GEGLU"}  # lora_diffusion/lora.py:53
TEXT_ENCODER_TARGETS = {"CLIPAttention"}  # lora_diffusion/lora.py:54


# ------------------------------------------------------------------------------------------------
# a1/a2/a3: the operator
# ------------------------------------------------------------------------------------------------
def lora_linear_forward(x, w, bias, down, up, scale):
    """lora_diffusion/lora.py:49-50:  linear(x) + lora_up(lora_down(x)) * scale."""
    return F.linear(x, w, bias) + F.linear(F.linear(x, down), up) * scale


def lora_linear_backward(x, w, down, up, scale, dy):
    """Autograd of lora.py:49-50 with W, b frozen (train_lora_dreambooth.py:595; lora.py:179-180):
    returns (dX, grad_down [r,K], grad_up [N,r]) for x [..,K], dy [..,N]."""
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    t = x2 @ down.t()            # [M,r]   lora_down(x)
    u = dy2 @ up                 # [M,r]   dY·B
    dx = dy2 @ w + scale * (u @ down)
    g_up = scale * (dy2.t() @ t)
    g_down = scale * (u.t() @ x2)
    return dx.reshape(x.shape), g_down, g_up


class LoraInjectedLinear(nn.Module):
    """CPU module with the reference's structure, init and forward (lora.py:32-50).  The class name matches
    on purpose: the finder and tune-scale logic key on it."""

    def __init__(self, in_features, out_features, bias=False, r=4):
        super().__init__()
        if r > min(in_features, out_features):  # lora.py:36-39
            raise ValueError(f"LoRA rank {r} must be less or equal than {min(in_features, out_features)}")
        self.linear = nn.Linear(in_features, out_features, bias)
        self.lora_down = nn.Linear(in_features, r, bias=False)
        self.lora_up = nn.Linear(r, out_features, bias=False)
        self.scale = 1.0
        nn.init.normal_(self.lora_down.weight, std=1 / r)  # lora.py:46
        nn.init.zeros_(self.lora_up.weight)  # lora.py:47

    def forward(self, x):
        return lora_linear_forward(x, self.linear.weight, self.linear.bias, self.lora_down.weight,
                                   self.lora_up.weight, self.scale)


# ------------------------------------------------------------------------------------------------
# a4/a5: enumeration and injection
# ------------------------------------------------------------------------------------------------
def find_targets(model, ancestor_names: Set[str], search=(nn.Linear,), exclude_parent=(LoraInjectedLinear,)):
    """lora.py:78-114: for each module whose class name is in `ancestor_names` (pre-order), each descendant
    that is an instance of `search` whose direct parent is not an `exclude_parent`.  Returns a list of
    (parent, attribute name, module, dotted path from the model root)."""
    paths = {id(m): n for n, m in model.named_modules()}
    out = []
    for anc in [m for m in model.modules() if m.__class__.__name__ in ancestor_names]:
        for full, mod in anc.named_modules():
            if not isinstance(mod, tuple(search)):
                continue
            *path, name = full.split(".")
            parent = anc
            for p in path:
                parent = parent.get_submodule(p)
            if exclude_parent and isinstance(parent, tuple(exclude_parent)):
                continue
            root = paths[id(anc)]
            out.append((parent, name, mod, (root + "." if root else "") + full))
    return out


def inject(model, targets: Set[str] = UNET_TARGETS, r: int = 4, factors: Optional[List[torch.Tensor]] = None):
    """lora.py:137-183.  Shares weight/bias Parameters (:164-166), follows the weight's device/dtype (:169),
    optional positional factors [up0, down0, ...] (:175-177).  Returns the flat parameter list
    [up0, down0, up1, ...] (what itertools.chain(*generators) yields) and the attribute names."""
    params, names = [], []
    for parent, name, child, _ in find_targets(model, targets):
        wrapped = LoraInjectedLinear(child.in_features, child.out_features, child.bias is not None, r)
        wrapped.linear.weight = child.weight
        if child.bias is not None:
            wrapped.linear.bias = child.bias
        wrapped.to(child.weight.device).to(child.weight.dtype)
        parent._modules[name] = wrapped
        if factors is not None:
            wrapped.lora_up.weight = nn.Parameter(factors.pop(0).clone())
            wrapped.lora_down.weight = nn.Parameter(factors.pop(0).clone())
        wrapped.lora_up.weight.requires_grad = True
        wrapped.lora_down.weight.requires_grad = True
        params += [wrapped.lora_up.weight, wrapped.lora_down.weight]
        names.append(name)
    return params, names


def extract_ups_downs(model, targets: Set[str] = UNET_TARGETS):
    """lora.py:186-198."""
    found = [(m.lora_up, m.lora_down) for _, _, m, _ in find_targets(model, targets, search=(LoraInjectedLinear,))]
    if not found:
        raise ValueError("No lora injected.")
    return found


# ------------------------------------------------------------------------------------------------
# a6/a7: merge, scale
# ------------------------------------------------------------------------------------------------
def merge_weight(w, up, down, alpha):
    """lora.py:410-424:  W + alpha * (up @ down).type(W.dtype)."""
    return w + alpha * (up @ down).type(w.dtype)


def tune_scale(model, alpha):
    """lora.py:597-600."""
    for m in model.modules():
        if m.__class__.__name__ == "LoraInjectedLinear":
            m.scale = alpha


# ------------------------------------------------------------------------------------------------
# a8/a9: losses
# ------------------------------------------------------------------------------------------------
def mse_loss(pred, target):
    """train_lora_dreambooth.py:875."""
    return F.mse_loss(pred.float(), target.float(), reduction="mean")


def prior_preservation_loss(pred, target, prior_loss_weight=1.0):
    """train_lora_dreambooth.py:855-873: halves of the batch = instance rows then class rows."""
    p_i, p_p = torch.chunk(pred, 2, dim=0)
    t_i, t_p = torch.chunk(target, 2, dim=0)
    inst = F.mse_loss(p_i.float(), t_i.float(), reduction="none").mean([1, 2, 3]).mean()
    prior = F.mse_loss(p_p.float(), t_p.float(), reduction="mean")
    return inst + prior_loss_weight * prior


def prepare_mask(mask, h, w):
    """cli_lora_pti.py:222-241: reshape to [B,1,8h,8w], nearest resize to [h,w], +0.05, divide by mean."""
    m = mask.reshape(mask.shape[0], 1, h * 8, w * 8)
    m = F.interpolate(m.float(), size=(h, w), mode="nearest") + 0.05
    return m / m.mean()


def masked_mse_loss(pred, target, mask):
    """cli_lora_pti.py:222-247."""
    m = prepare_mask(mask, pred.shape[-2], pred.shape[-1])
    return F.mse_loss((pred * m).float(), (target * m).float(), reduction="mean")


# ------------------------------------------------------------------------------------------------
# a10: DDPM schedule (diffusers DDPMScheduler is not vendored: restated from the DDPM definitions with
# the public SD schedule — scaled-linear betas 0.00085→0.012, 1000 steps.  Parity unpinned.)
# ------------------------------------------------------------------------------------------------
def ddpm_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(x0, noise, t, acp):
    """train_lora_dreambooth.py:837 → sqrt(ᾱ_t)·x0 + sqrt(1-ᾱ_t)·ε."""
    a = acp[t].sqrt().reshape(-1, 1, 1, 1)
    s = (1 - acp[t]).sqrt().reshape(-1, 1, 1, 1)
    return a * x0 + s * noise


def get_velocity(x0, noise, t, acp):
    """train_lora_dreambooth.py:849 → sqrt(ᾱ_t)·ε − sqrt(1-ᾱ_t)·x0."""
    a = acp[t].sqrt().reshape(-1, 1, 1, 1)
    s = (1 - acp[t]).sqrt().reshape(-1, 1, 1, 1)
    return a * noise - s * x0


# ------------------------------------------------------------------------------------------------
# row H / f-2: clip + AdamW, restated op by op
# ------------------------------------------------------------------------------------------------
def clip_grad_norm(grads: Sequence[torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (train_lora_dreambooth.py:884): total L2 norm over all grads,
    coefficient max_norm/(norm+1e-6) clamped to 1, applied in place.  Returns the total norm."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    """torch.optim.AdamW single-tensor update (train_lora_dreambooth.py:659-676,885), in place."""
    p.mul_(1 - lr * weight_decay)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def linear_schedule_factor(epoch: int, num_warmup_steps: int, num_training_steps: int) -> float:
    """λ(epoch) of get_scheduler("linear", …) — cli_lora_pti.py:746-751 (default `lr_scheduler_lora="linear"`, 0 warm-up
    steps, :534-535), train_lora_dreambooth.py:737-743 with --lr_scheduler linear.  diffusers' function (third party, not in
    the reference tree): a torch LambdaLR with a linear ramp over the warm-up steps and a linear decay to 0 at
    num_training_steps — restated from its published definition, parity unpinned for the formula itself."""
    if epoch < num_warmup_steps:
        return float(epoch) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - epoch) / float(max(1, num_training_steps - num_warmup_steps)))


# ------------------------------------------------------------------------------------------------
# row H: the step harness (train_lora_dreambooth.py:811-888), synthetic latents instead of VAE/CLIP
# ------------------------------------------------------------------------------------------------
def synthetic_batch(step: int, batch: int, latent_hw: int, ctx_len: int, ctx_dim: int, seed_base: int = 1000,
                    t_max: int = 1000):
    """Per-step synthetic inputs, identical on every rank (set_seed semantics, train_lora_dreambooth.py:509-510)
    and on CPU/GPU (generated on the host, copied).  t_max: exclusive bound of the timestep draw (the PTI loop draws below
    int(1000 · t_mutliplier), cli_lora_pti.py:190-195)."""
    g = torch.Generator().manual_seed(seed_base + step)
    latents = torch.randn(batch, 4, latent_hw, latent_hw, generator=g) * 0.18215
    noise = torch.randn(batch, 4, latent_hw, latent_hw, generator=g)
    t = torch.randint(0, t_max, (batch,), generator=g)
    ctx = torch.randn(batch, ctx_len, ctx_dim, generator=g)
    return latents, noise, t, ctx


def train_steps(unet, params: List[torch.Tensor], steps: int, batch: int, latent_hw: int, ctx_len: int,
                ctx_dim: int, lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0, with_prior=False,
                prior_loss_weight=1.0, v_prediction=False, world: int = 1, first_step: int = 0,
                state: Optional[dict] = None, lr_schedule=None):
    """`steps` optimizer steps of the reference loop on CPU.  `world` > 1 emulates synchronous data
    parallelism: each virtual rank takes its own slice of a `world*batch` batch and the gradients are
    averaged (DDP mean all-reduce, train_lora_dreambooth.py:744-757,877).  Returns the loss history.
    `state`: a dict that carries the optimizer state (Adam moments, step count) from one call to the next, so that a
    trajectory can be produced in pieces (a test that looks at the gradients of the first step, then continues).
    `lr_schedule`: λ(epoch) on the learning rate; this loop calls lr_scheduler.step() AFTER optimizer.step()
    (train_lora_dreambooth.py:885-886), so optimizer step k (from 0) runs at lr·λ(k)."""
    acp = ddpm_alphas_cumprod()
    if state is not None and "m" in state:
        m, v, done = state["m"], state["v"], state["t"]
    else:
        m = [torch.zeros_like(p) for p in params]
        v = [torch.zeros_like(p) for p in params]
        done = 0
    losses = []
    for s in range(first_step, first_step + steps):
        latents, noise, t, ctx = synthetic_batch(s, batch * world, latent_hw, ctx_len, ctx_dim)
        for p in params:
            p.grad = None
        step_losses = []
        for rk in range(world):
            sl = slice(rk * batch, (rk + 1) * batch)
            noisy = add_noise(latents[sl], noise[sl], t[sl], acp)
            pred = unet(noisy, t[sl], ctx[sl]).sample
            target = get_velocity(latents[sl], noise[sl], t[sl], acp) if v_prediction else noise[sl]
            loss = prior_preservation_loss(pred, target, prior_loss_weight) if with_prior else mse_loss(pred, target)
            (loss / world).backward()  # mean over ranks == DDP's averaged all-reduce
            step_losses.append(loss.item())
        grads = [p.grad for p in params]
        clip_grad_norm(grads, max_grad_norm)
        for p, g, mm, vv in zip(params, grads, m, v):
            with torch.no_grad():
                adamw_step(p, g, mm, vv, done + s - first_step + 1,
                           lr * (lr_schedule(done + s - first_step) if lr_schedule is not None else 1.0),
                           weight_decay=weight_decay)
        losses.append(sum(step_losses) / world)
    if state is not None:
        state.update(m=m, v=v, t=done + steps)
    return losses


# ------------------------------------------------------------------------------------------------
# config 5: the PTI tuning phase with continue_inversion (cli_lora_pti.py:408-451, loss_step :170-248, set-up :693-738)
# ------------------------------------------------------------------------------------------------
def freeze_all_but_token_embeddings(text_encoder) -> nn.Parameter:
    """cli_lora_pti.py:704-722: requires_grad_(False); under continue_inversion requires_grad_(True) and then the encoder
    layers, the final layer norm and the position embedding frozen again — what is left trainable is the token table
    (`get_input_embeddings().parameters()`, the optimizer group of :708-716).  Returns that Parameter."""
    text_encoder.requires_grad_(False)
    table = text_encoder.get_input_embeddings().weight
    table.requires_grad_(True)
    return table


def synthetic_token_ids(step: int, batch: int, ctx_len: int, vocab: int, seed_base: int = 7000, bos: int = 1, eos: int = 2):
    """Caption-shaped ids: bos, a few words, then eos repeated to the end (the tokenizer pads with one token: every padding
    position hits the SAME table row, so the table gradient sums many positions per token)."""
    g = torch.Generator().manual_seed(seed_base + step)
    ids = torch.randint(3, vocab, (batch, ctx_len), generator=g)
    n_words = torch.randint(1, max(2, ctx_len - 2), (batch,), generator=g)
    ids[:, 0] = bos
    for b in range(batch):
        ids[b, 1 + int(n_words[b]):] = eos
    return ids


def pti_tuning_steps(unet, text_encoder, lora_params: List[torch.Tensor], steps: int, batch: int, latent_hw: int,
                     ctx_len: int, vocab: int, lr_unet=1e-4, lr_embed=5e-4, weight_decay=1e-3, max_grad_norm=1.0,
                     v_prediction=True, t_multiplier=0.8, masks: Optional[Sequence[torch.Tensor]] = None, first_step: int = 0,
                     bos: int = 1, eos: int = 2, lr_schedule=None):
    """`steps` iterations of perform_tuning (cli_lora_pti.py:424-451) with the optimizer of :738 —
    AdamW([{unet LoRA, lr_unet}, {token table, continue_inversion_lr | ti_lr}], weight_decay=weight_decay_lora) — on synthetic
    latents: loss_step's draw `randint(0, int(1000·t_mutliplier))` (:190-195, 0.8 at :444), add_noise, the text encoder INSIDE
    the step (:199-206), ε- or v-target (:215-220), optional mask (:222-245), mse (:247); backward; clip_grad_norm_ over
    chain(unet.parameters(), text_encoder.parameters()) (:448-450); step.  fp32 (the reference autocasts; the parity target
    is the fp32 arithmetic).  Returns the loss history; `lora_params` and the token table are updated in place.
    `lr_schedule`: λ(epoch); perform_tuning calls `lr_scheduler_lora.step()` BEFORE every batch (:434), so iteration k (from 0)
    runs at lr·λ(k + 1) — with the default linear schedule (:534-535,746-751) the first step is already at lr·(1 − 1/N)."""
    acp = ddpm_alphas_cumprod()
    table = text_encoder.get_input_embeddings().weight
    params = list(lora_params) + [table]
    lrs = [lr_unet] * len(lora_params) + [lr_embed]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    ctx_dim = table.shape[1]
    losses = []
    for s in range(first_step, first_step + steps):
        latents, noise, t, _ = synthetic_batch(s, batch, latent_hw, ctx_len, ctx_dim, t_max=int(1000 * t_multiplier))
        ids = synthetic_token_ids(s, batch, ctx_len, vocab, bos=bos, eos=eos)
        for p in params:
            p.grad = None
        noisy = add_noise(latents, noise, t, acp)
        ehs = text_encoder(ids)[0]
        pred = unet(noisy, t, ehs).sample
        target = get_velocity(latents, noise, t, acp) if v_prediction else noise
        loss = masked_mse_loss(pred, target, masks[s - first_step]) if masks is not None else mse_loss(pred, target)
        loss.backward()
        grads = [p.grad for p in params]
        clip_grad_norm(grads, max_grad_norm)
        for p, g, mm, vv, lr in zip(params, grads, m, v, lrs):
            with torch.no_grad():
                adamw_step(p, g, mm, vv, s - first_step + 1,
                           lr * (lr_schedule(s - first_step + 1) if lr_schedule is not None else 1.0), weight_decay=weight_decay)
        losses.append(loss.item())
    return losses


def flat_params(params: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([p.detach().reshape(-1) for p in params])
