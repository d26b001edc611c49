"""The reference's attention hook, backed by the HIP short-context attention core.

The reference switches the attention implementation of a model with
`set_use_memory_efficient_attention_xformers(module, valid)` (lora_diffusion/xformers_utils.py:41-70, called by
training_scripts/train_lora_dreambooth.py:623-625 under `--use_xformers`).  This module exports the same function
with the same signature.  Instead of xformers it installs, on every attention module under `module`, a forward that
runs `softmax(QKᵀ·scale)V` through the HIP attention cores — csrc/attn_ctx.hip for a short key/value sequence (the
cross-attention over the text tokens), csrc/attn_flash.hip for a long one (self-attention over thousands of tokens) —
directly on the [B, T, H·d] tensors its `to_q/to_k/to_v` linears (the LoRA targets, lora_diffusion/lora.py:53)
produce.  Everything the kernels do not cover (fp32 tensors, head dims above 160, masks, CPU tensors, exotic module
options) is handed back, untouched, to the module's own forward: the product implements what it accelerates and nothing
else.

Attention modules are recognised structurally, like the reference recognises LoRA targets by class name
(lora.py:78-114): class name "CrossAttention" or "Attention" with `heads`, `to_q`, `to_k`, `to_v` and a `to_out`
sequence — the layout of diffusers' class in every version the reference supports, and of the build's harness UNet.
"""
import functools
import torch
from torch import nn

from . import _native as nat
from ._fastattr import factor_weights, linear_params
from .groups import ctx_cross_attention, qkv_self_attention
from .ops import feed_forward_geglu, lora_tail
from .sandwich import ctx_attention, flash_attention

ATTENTION_CLASS_NAMES = {"CrossAttention", "Attention"}
_ORIG = "_dfa_original_forward"


def _is_attention_module(m: nn.Module) -> bool:
    if m.__class__.__name__ not in ATTENTION_CLASS_NAMES:
        return False
    if not all(hasattr(m, a) for a in ("heads", "to_q", "to_k", "to_v", "to_out")):
        return False
    # options whose arithmetic the short-context kernel does not reproduce: leave such modules alone
    for attr in ("added_kv_proj_dim", "group_norm", "norm_cross", "spatial_norm"):
        if getattr(m, attr, None) not in (None, False):
            return False
    return True


def _out_features(linear: nn.Module) -> int:
    inner = getattr(linear, "linear", linear)  # LoraInjectedLinear keeps the nn.Linear in `.linear` (lora.py:42)
    return inner.out_features


def _compute_dtype(x: torch.Tensor) -> torch.dtype:
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return x.dtype


def _hip_forward(self, hidden_states, *args, **kwargs):
    """Replacement forward.  Understands both generations of the diffusers signature — (hidden_states, context, mask)
    and (hidden_states, encoder_hidden_states=..., attention_mask=...) — and passes anything else through."""
    original = self.__dict__[_ORIG]
    ctx = kwargs.get("encoder_hidden_states", kwargs.get("context", args[0] if len(args) > 0 else None))
    mask = kwargs.get("attention_mask", kwargs.get("mask", args[1] if len(args) > 1 else None))
    unknown = len(args) > 2 or any(k not in ("encoder_hidden_states", "context", "attention_mask", "mask") for k in kwargs)
    heads = int(self.heads)
    self_attention = ctx is None
    if self_attention:
        ctx = hidden_states
    core = None
    cdtype = _compute_dtype(hidden_states)
    to_q = self.to_q
    width = _out_features(to_q)
    if (not unknown and mask is None and hidden_states.is_cuda and hidden_states.dim() == 3 and ctx.dim() == 3
            and width % heads == 0):
        shape = (hidden_states.shape[0], hidden_states.shape[1], ctx.shape[1], heads, width // heads, cdtype)
        if nat.attn_ctx_supported(*shape):      # up to 128 keys: the whole K/V of a head in LDS, one tile
            core = ctx_attention
        elif nat.attn_flash_supported(*shape):  # any length: online softmax over key tiles
            core = flash_attention
    if core is None:
        return original(hidden_states, *args, **kwargs)
    scale = getattr(self, "scale", None)
    scale = float(scale) if isinstance(scale, (int, float)) else None
    # Grouped LoRA projections (set up by trainer.LoraSlab.enable_groups): q/k/v of a self-attention in one launch each
    # way, K/V of all cross-attentions over the same context in one launch per pass — the cores then work on column
    # slices of the shared buffers (groups.py).
    qkv = self.__dict__.get("_dfa_qkv")
    kvg = self.__dict__.get("_dfa_ctx")
    layers = list(self.to_out)  # linear (LoRA target), dropout
    if self_attention and qkv is not None and qkv.usable(hidden_states, cdtype) and nat.attn_flash_supported(*shape):
        # (the grouped paths run `to_out[0]` inside the attention core's autograd node when it is a plain wrapped layer)
        tail = lora_tail(layers[0], cdtype) if layers else None
        out = qkv_self_attention(qkv, hidden_states, heads, scale, cdtype, tail)
        if tail is not None:
            del layers[0]
    elif not self_attention and kvg is not None and core is ctx_attention and kvg[0].usable(ctx, cdtype):
        tail = lora_tail(layers[0], cdtype) if layers else None
        out = ctx_cross_attention(kvg[0], kvg[1], to_q(hidden_states), ctx, heads, scale, cdtype, tail)
        if tail is not None:
            del layers[0]
    else:
        q, k, v = to_q(hidden_states), self.to_k(ctx), self.to_v(ctx)
        if q.dtype != k.dtype:  # mixed module dtypes outside autocast: compute in the query's dtype
            k, v = k.to(q.dtype), v.to(q.dtype)
        out = core(q, k, v, heads, scale)
    for layer in layers:
        if type(layer) is nn.Dropout and layer.p == 0.0 and not layer._forward_hooks and not layer._forward_pre_hooks:
            continue  # (the identity — Stable Diffusion's attention blocks are built with dropout 0.0 — without a module call)
        out = layer(out)
    return out


def _attach_dropin_groups(module: nn.Module, valid: bool) -> None:
    """Grouped LoRA projections for a model that is NOT under a trainer.LoraSlab — what an unchanged reference trainer gets
    from flipping its one switch (train_lora_dreambooth.py:623-625): to_q/to_k/to_v of every self-attention as one launch each
    way, to_k/to_v of all cross-attentions over the same context as one launch per pass (groups.py).  The members' packed
    operands live in the ops.PackRegistry `inject_trainable_lora` made for them; their factor gradients go to the drop-in
    sink.  A LoraTrainer built later replaces these groups with its own."""
    from .core import LoraInjectedLinear
    from .groups import CtxKVGroup, QKVGroup

    mods = [m for m in module.modules() if _is_attention_module(m)]
    regs = set()
    for m in mods:
        for key in ("_dfa_qkv", "_dfa_ctx"):
            g = m.__dict__.get(key)
            g = g[0] if isinstance(g, tuple) else g
            if g is not None and g.sinks is None:  # only groups this function made (a trainer's have sinks)
                regs.add(g.registry)
                m.__dict__.pop(key)
    for reg in regs:
        if reg is not None:
            reg.drop_groups()
    hook = module.__dict__.pop("_dfa_ctx_pass_hook", None)
    if hook is not None:
        hook.remove()
    if not valid:
        return
    cross = {}
    for name, m in module.named_modules():
        if not _is_attention_module(m) or "_dfa_qkv" in m.__dict__ or "_dfa_ctx" in m.__dict__:
            continue
        trio = [m.to_q, m.to_k, m.to_v]
        if not all(type(l) is LoraInjectedLinear and "_dfa_grad_sink" not in l.__dict__ for l in trio):
            continue
        reg = trio[0].__dict__.get("_dfa_packreg")
        if reg is None or any(l.__dict__.get("_dfa_packreg") is not reg for l in trio):
            continue
        is_cross = (m.to_k.linear.in_features != m.to_q.linear.in_features) or name.split(".")[-1] == "attn2"
        if is_cross:
            cross.setdefault((id(reg), m.to_k.linear.in_features, m.to_k.lora_down.weight.shape[0]), (reg, []))[1].append(m)
        elif QKVGroup.eligible(trio):
            grp = QKVGroup(trio, None)
            if reg.add_group(grp):
                m.__dict__["_dfa_qkv"] = grp
    groups = []
    for reg, members in cross.values():
        layers = [l for m in members for l in (m.to_k, m.to_v)]
        if len(members) < 2 or not CtxKVGroup.eligible(layers):
            continue
        grp = CtxKVGroup(members, layers, None)
        if reg.add_group(grp):
            for i, m in enumerate(members):
                m.__dict__["_dfa_ctx"] = (grp, i)
            groups.append(grp)
    if groups:  # a new forward pass of the model starts a new shared K/V projection
        module.__dict__["_dfa_ctx_pass_hook"] = module.register_forward_pre_hook(
            lambda mod, args, gs=tuple(groups): [g.new_pass() for g in gs] and None)


def set_use_hip_attention(module: nn.Module, valid: bool = True) -> int:
    """Install (valid=True) or remove (valid=False) the HIP short-context attention forward on every attention module
    under `module`.  Returns the number of modules touched.  Idempotent."""
    import os

    if os.environ.get("DFA_DROPIN_GROUPS", "1") != "0":
        _attach_dropin_groups(module, bool(valid))
    touched = 0
    for m in module.modules():
        if not _is_attention_module(m):
            continue
        has = _ORIG in m.__dict__
        if valid and not has:
            m.__dict__[_ORIG] = m.forward  # the bound method (class forward, or whatever was installed before)
            m.forward = functools.partial(_hip_forward, m)
            touched += 1
        elif not valid and has:
            orig = m.__dict__.pop(_ORIG)
            del m.forward  # drop the instance attribute: the class's forward is visible again
            if getattr(orig, "__func__", None) is not type(m).forward:
                m.forward = orig  # something else had been installed before us: put it back
            touched += 1
    return touched


def _hip_geglu_forward(self, hidden_states, *args, **kwargs):
    """diffusers GEGLU.forward: `hidden, gate = proj(x).chunk(2, -1); hidden * gelu(gate)` — here one fused pass
    after the `proj` LoRA linear (and one in backward)."""
    if args or kwargs or not hidden_states.is_cuda:
        return self.__dict__[_ORIG](hidden_states, *args, **kwargs)
    from .sandwich import geglu_gate

    proj = self.proj
    from .core import LoraInjectedLinear

    plain = (type(proj) is LoraInjectedLinear and "forward" not in proj.__dict__ and not proj._forward_hooks
             and not proj._forward_pre_hooks)
    if plain:
        from .ops import lora_linear_geglu

        return lora_linear_geglu(proj, hidden_states)  # the gate rides in the LoRA GEMM's epilogue: one launch
    return geglu_gate(proj(hidden_states))


def _plain_lora_linear(m) -> bool:
    from .core import LoraInjectedLinear

    return (type(m) is LoraInjectedLinear and "forward" not in m.__dict__ and not m._forward_hooks
            and not m._forward_pre_hooks)


def _is_geglu_feed_forward(m: nn.Module) -> bool:
    """diffusers FeedForward with a GEGLU activation: `net = [GEGLU(proj), Dropout, Linear]`, run as `for f in net: x = f(x)`."""
    net = getattr(m, "net", None)
    if m.__class__.__name__ != "FeedForward" or net is None or len(net) != 3:
        return False
    return (net[0].__class__.__name__ == "GEGLU" and hasattr(net[0], "proj") and isinstance(net[1], nn.Dropout)
            and type(net[2]) is nn.Linear)


def _hip_feed_forward(self, hidden_states, *args, **kwargs):
    """FeedForward.forward with the gate inside GEMM epilogues in BOTH directions (ops.feed_forward_geglu): forward in the
    `proj` LoRA launch, backward in the launch that computes the second linear layer's input gradient.  Anything outside
    that envelope — extra arguments, active dropout, a `proj` that is not a plain LoraInjectedLinear, a second layer that is
    trainable or hooked, no gradient wanted — takes the module's own forward (whose GEGLU keeps its forward-only fusion)."""
    geglu, drop, lin2 = self.net  # (a ModuleList of three: _is_geglu_feed_forward)
    proj = geglu.proj
    w2, b2 = linear_params(lin2)
    ok = (not args and not kwargs and hidden_states.is_cuda and torch.is_grad_enabled()
          and not (drop.training and drop.p > 0) and _plain_lora_linear(proj)
          and "forward" not in lin2.__dict__ and not lin2._forward_hooks and not lin2._forward_pre_hooks
          and not w2.requires_grad and (b2 is None or not b2.requires_grad)
          and (hidden_states.requires_grad or factor_weights(proj)[1].requires_grad))
    if not ok:
        return self.__dict__[_ORIG](hidden_states, *args, **kwargs)
    return feed_forward_geglu(proj, lin2, hidden_states)


def set_use_hip_geglu(module: nn.Module, valid: bool = True) -> int:
    """Install / remove the fused GEGLU gate on every module whose class is named "GEGLU" (the LoRA target class of
    lora_diffusion/lora.py:53) and that has a `proj` linear — and, on every "FeedForward" module built around such a GEGLU,
    the forward that also folds the gate's BACKWARD into the following linear layer's backward GEMM.  The reference has no
    switch for this op; a trainer that wants it adds one line next to its `--use_xformers` line.  Returns the number of GEGLU
    modules touched."""
    import os

    ff_fusion = os.environ.get("DFA_NO_FF_FUSION", "0") != "1"  # dev knob (tools/ab_env.sh): forward-only gate fusion
    for m in module.modules():
        if not _is_geglu_feed_forward(m):
            continue
        has = _ORIG in m.__dict__
        if valid and not has and ff_fusion:
            m.__dict__[_ORIG] = m.forward
            m.forward = functools.partial(_hip_feed_forward, m)
        elif not valid and has:
            orig = m.__dict__.pop(_ORIG)
            del m.forward
            if getattr(orig, "__func__", None) is not type(m).forward:
                m.forward = orig
    touched = 0
    for m in module.modules():
        if m.__class__.__name__ != "GEGLU" or not hasattr(m, "proj"):
            continue
        has = _ORIG in m.__dict__
        if valid and not has:
            m.__dict__[_ORIG] = m.forward
            m.forward = functools.partial(_hip_geglu_forward, m)
            touched += 1
        elif not valid and has:
            orig = m.__dict__.pop(_ORIG)
            del m.forward
            if getattr(orig, "__func__", None) is not type(m).forward:
                m.forward = orig
            touched += 1
    return touched


def set_use_memory_efficient_attention_xformers(module: nn.Module, valid: bool) -> None:
    """Same name and signature as the reference's hook (lora_diffusion/xformers_utils.py:41-70), so the trainers'
    `--use_xformers` path (train_lora_dreambooth.py:623-625) switches the HIP attention cores on without edits.
    Modules or calls outside the kernels' envelope (e.g. the VAE's 512-wide single head) keep deferring to their own
    forward, like the reference turns xformers off per block when its probe fails (xformers_utils.py:46-60)."""
    set_use_hip_attention(module, bool(valid))
    # The switch a reference trainer flips is the only hook an UNCHANGED trainer ever calls, so it also turns on the fused
    # GEGLU gate of the feed-forward blocks under `module` (set_use_hip_geglu: same arithmetic as diffusers' GEGLU.forward /
    # FeedForward.forward, inside the `proj` LoRA launch and the ff.net.2 backward launch); modules without a GEGLU — the
    # VAE the trainers also pass here — are untouched.  DFA_XFORMERS_SWITCH_GEGLU=0 keeps the two switches separate.
    import os

    if os.environ.get("DFA_XFORMERS_SWITCH_GEGLU", "1") != "0":
        set_use_hip_geglu(module, bool(valid))


def test_xformers_backwards(size: int) -> bool:
    """The reference probes whether its attention backend can differentiate a head size (xformers_utils.py:17-38).
    Here: whether the HIP attention kernels cover that head size in f16."""
    return bool(nat.attn_flash_supported(1, 64, 64, 1, int(size), torch.float16))
