"""Core of the `lora_diffusion` API on the MI355X-native path: the LoRA module, the target finders (whose
enumeration order is the on-disk index), injection, merging and the monkeypatch family.

Reference: lora_diffusion/lora.py:32-198 and :410-600 — same names, signatures, side effects (caller lists are
consumed with pop(0), module trees are edited in place) and errors.  What differs is underneath:
`LoraInjectedLinear.forward` is one fused HIP kernel (ops.lora_linear) and `weight_apply_lora` a HIP merge kernel.
"""
from typing import List, Optional, Set, Type, Union

import torch
import torch.nn as nn

from . import _native as nat
from .ops import lora_linear


class LoraInjectedLinear(nn.Module):
    """Reference: lora_diffusion/lora.py:32-50.  Same sub-modules (`linear`, `lora_down`, `lora_up`), same
    `scale` attribute, same init (down ~ N(0, (1/r)²), up = 0), same ValueError for an over-large rank; the
    class name is load-bearing (`tune_lora_scale` / `inspect_lora` match on it)."""

    def __init__(self, in_features, out_features, bias=False, r=4):
        super().__init__()
        limit = min(in_features, out_features)
        if r > limit:
            raise ValueError(f"LoRA rank {r} must be less or equal than {limit}")
        self.linear = nn.Linear(in_features, out_features, bias)
        self.lora_down = nn.Linear(in_features, r, bias=False)
        self.lora_up = nn.Linear(r, out_features, bias=False)
        self.scale = 1.0
        nn.init.normal_(self.lora_down.weight, std=1 / r)
        nn.init.zeros_(self.lora_up.weight)

    def forward(self, input):
        # y = W x + b + scale · up(down(x)) — fused on the HIP device (csrc/lora_gemm.hip)
        return lora_linear(self, input)


UNET_DEFAULT_TARGET_REPLACE = {"CrossAttention", "Attention", "GEGLU"}


TEXT_ENCODER_DEFAULT_TARGET_REPLACE = {"CLIPAttention"}


DEFAULT_TARGET_REPLACE = UNET_DEFAULT_TARGET_REPLACE


EMBED_FLAG = "<embed>"


def _matches(module, classes) -> bool:
    return any(isinstance(module, c) for c in classes)


def _find_children(model, search_class: List[Type[nn.Module]] = [nn.Linear]):
    """(parent, name, child) for every direct child of any module that is an instance of `search_class`
    (reference: lora.py:61-75)."""
    for parent in model.modules():
        for name, child in parent.named_children():
            if _matches(child, search_class):
                yield parent, name, child


def _find_modules_v2(
    model,
    ancestor_class: Set[str] = DEFAULT_TARGET_REPLACE,
    search_class: List[Type[nn.Module]] = [nn.Linear],
    exclude_children_of: Optional[List[Type[nn.Module]]] = [LoraInjectedLinear],
):
    """(parent, name, module) for every `search_class` descendant of a module whose class NAME is in
    `ancestor_class`, skipping direct children of `exclude_children_of` (reference: lora.py:78-114).
    Lazy like the reference: the tree may be edited between yields."""
    for ancestor in (m for m in model.modules() if type(m).__name__ in ancestor_class):
        for path, module in ancestor.named_modules():
            if not _matches(module, search_class):
                continue
            *parents, name = path.split(".")
            holder = ancestor
            for step in parents:
                holder = holder.get_submodule(step)
            if exclude_children_of and _matches(holder, exclude_children_of):
                continue
            yield holder, name, module


def _find_modules_old(
    model,
    ancestor_class: Set[str] = DEFAULT_TARGET_REPLACE,
    search_class: List[Type[nn.Module]] = [nn.Linear],
    exclude_children_of: Optional[List[Type[nn.Module]]] = [LoraInjectedLinear],
):
    """Legacy finder kept for API completeness (reference: lora.py:117-131): exact class match, ancestor
    returned as the holder, result printed."""
    found = [
        (anc, name, mod)
        for anc in model.modules()
        if type(anc).__name__ in ancestor_class
        for name, mod in anc.named_modules()
        if type(mod) in search_class
    ]
    print(found)
    return found


_find_modules = _find_modules_v2


def _wrap_linear(holder, name, source, r, up=None, down=None, follow_weight=True):
    """Replaces holder.<name> by a LoraInjectedLinear that SHARES source.weight / source.bias (the same
    Parameter objects, reference lora.py:164-166).  Optional factor tensors are installed as Parameters in the
    base weight's dtype."""
    weight, bias = source.weight, source.bias
    wrapped = LoraInjectedLinear(source.in_features, source.out_features, bias is not None, r)
    wrapped.linear.weight = weight
    if bias is not None:
        wrapped.linear.bias = bias
    holder._modules[name] = wrapped
    if up is not None:
        wrapped.lora_up.weight = nn.Parameter(up.type(weight.dtype))
        wrapped.lora_down.weight = nn.Parameter(down.type(weight.dtype))
    if follow_weight:
        wrapped.to(weight.device)
    return wrapped


def inject_trainable_lora(
    model: nn.Module,
    target_replace_module: Set[str] = DEFAULT_TARGET_REPLACE,
    r: int = 4,
    loras=None,  # path to lora .pt
):
    """Wraps every target nn.Linear and returns ([up.parameters(), down.parameters(), ...], names)
    (reference: lora.py:137-183).  `loras` is a path to a positional `[up0, down0, up1, ...]` .pt list; unlike
    the reference (which raises TypeError on its own files) plain tensors are accepted and cast."""
    params, names, wrapped_modules = [], [], []
    if loras is not None:
        loras = torch.load(loras, map_location="cpu", weights_only=True)
    for holder, name, child in _find_modules(model, target_replace_module, search_class=[nn.Linear]):
        wrapped = _wrap_linear(holder, name, child, r, follow_weight=False)
        wrapped_modules.append(wrapped)
        wrapped.to(child.weight.device).to(child.weight.dtype)
        if loras is not None:
            up, down = loras.pop(0), loras.pop(0)
            like = wrapped.lora_up.weight
            wrapped.lora_up.weight = nn.Parameter(up.detach().to(like.device, like.dtype))
            wrapped.lora_down.weight = nn.Parameter(down.detach().to(like.device, like.dtype))
        params.append(wrapped.lora_up.parameters())
        params.append(wrapped.lora_down.parameters())
        wrapped.lora_up.weight.requires_grad = True
        wrapped.lora_down.weight.requires_grad = True
        names.append(name)
    from .ops import register_pack_group

    register_pack_group(wrapped_modules)  # drop-in mode: one pack launch per parameter update for all of them (ops.PackRegistry)
    return params, names


def extract_lora_ups_down(model, target_replace_module=DEFAULT_TARGET_REPLACE):
    """[(lora_up, lora_down), ...] in enumeration order; ValueError when nothing is injected (lora.py:186-198)."""
    pairs = [
        (m.lora_up, m.lora_down)
        for _, _, m in _find_modules(model, target_replace_module, search_class=[LoraInjectedLinear])
    ]
    if not pairs:
        raise ValueError("No lora injected.")
    return pairs


def weight_apply_lora(model, loras, target_replace_module=DEFAULT_TARGET_REPLACE, alpha=1.0):
    """W ← W + α·(up @ down).type(W.dtype) for every target nn.Linear, as a NEW Parameter on the weight's own device
    (lora.py:410-424).  All layers are merged by ONE HIP launch (csrc/optim.hip: merge_batched_kernel).  Weights that
    live on the host — the reference's `lora_add` moves the pipeline to the CPU first, cli_lora_add.py:74-78 — are
    staged through the HIP device and handed back; nothing is computed on the CPU."""
    targets = [child for _, _, child in _find_modules(model, target_replace_module, search_class=[nn.Linear])]
    if not targets:
        return
    device = nat.staging_device(*(c.weight for c in targets))  # raises before `loras` is touched
    entries = []
    for child in targets:
        weight = child.weight
        up, down = loras.pop(0), loras.pop(0)
        merged = weight.detach().to(device, copy=True).contiguous()
        held = up.dtype if up.dtype in (torch.float16, torch.bfloat16) else torch.float32
        entries.append((merged, down.detach().to(device).float().contiguous(),
                        up.detach().to(device).float().contiguous(), held))
    nat.lora_merge_weight_batched(entries, alpha)
    for child, entry in zip(targets, entries):
        child.weight = nn.Parameter(entry[0].to(child.weight.device))


def monkeypatch_lora(model, loras, target_replace_module=DEFAULT_TARGET_REPLACE, r: int = 4):
    """Wrap plain Linears and install given factors (lora.py:427-459)."""
    for holder, name, child in _find_modules(model, target_replace_module, search_class=[nn.Linear]):
        _wrap_linear(holder, name, child, r, up=loras.pop(0), down=loras.pop(0))


def monkeypatch_replace_lora(model, loras, target_replace_module=DEFAULT_TARGET_REPLACE, r: int = 4):
    """Replace the factors of already wrapped layers, possibly with a new rank (lora.py:462-494)."""
    for holder, name, child in _find_modules(model, target_replace_module, search_class=[LoraInjectedLinear]):
        _wrap_linear(holder, name, child.linear, r, up=loras.pop(0), down=loras.pop(0))


def monkeypatch_or_replace_lora(
    model,
    loras,
    target_replace_module=DEFAULT_TARGET_REPLACE,
    r: Union[int, List[int]] = 4,
):
    """Wrap or re-wrap; `r` may be a per-layer list consumed with pop(0) (lora.py:497-538)."""
    for holder, name, child in _find_modules(
        model, target_replace_module, search_class=[nn.Linear, LoraInjectedLinear]
    ):
        source = child.linear if isinstance(child, LoraInjectedLinear) else child
        rank = r.pop(0) if isinstance(r, list) else r
        _wrap_linear(holder, name, source, rank, up=loras.pop(0), down=loras.pop(0))


def monkeypatch_remove_lora(model):
    """Every LoraInjectedLinear child becomes a plain nn.Linear again, sharing W/b (lora.py:554-567)."""
    for holder, name, child in _find_children(model, search_class=[LoraInjectedLinear]):
        src = child.linear
        plain = nn.Linear(src.in_features, src.out_features, src.bias is not None)
        plain.weight = src.weight
        if src.bias is not None:
            plain.bias = src.bias
        holder._modules[name] = plain


def monkeypatch_add_lora(
    model,
    loras,
    target_replace_module=DEFAULT_TARGET_REPLACE,
    alpha: float = 1.0,
    beta: float = 1.0,
):
    """factor ← α·given + β·current for up and down separately (lora.py:570-594)."""
    for holder, name, child in _find_modules(model, target_replace_module, search_class=[LoraInjectedLinear]):
        weight = child.linear.weight
        layer = holder._modules[name]
        for attr in ("lora_up", "lora_down"):
            given = loras.pop(0).type(weight.dtype).to(weight.device)
            current = getattr(layer, attr).weight.to(weight.device)
            getattr(layer, attr).weight = nn.Parameter(given * alpha + current * beta)
        layer.to(weight.device)


def tune_lora_scale(model, alpha: float = 1.0):
    """Sets `.scale` on every module whose class NAME is LoraInjectedLinear (lora.py:597-600)."""
    for module in model.modules():
        if type(module).__name__ == "LoraInjectedLinear":
            module.scale = alpha
