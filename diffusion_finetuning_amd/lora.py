"""`lora_diffusion.lora` API surface on the MI355X-native hot path.

Every public name of the reference's lora_diffusion/lora.py exists here with the same signature, argument
meaning, side effects (caller lists are consumed with pop(0), module trees are edited in place) and error
behaviour, so training_scripts/train_lora_dreambooth.py and lora_diffusion/cli_lora_pti.py run unchanged.
What differs is underneath: `LoraInjectedLinear.forward` is one fused HIP kernel (ops.lora_linear) and
`weight_apply_lora` is a HIP merge kernel.  Reference line numbers are cited per function.
"""
import json
from itertools import groupby
from typing import Dict, List, Optional, Set, Tuple, Type, Union

import torch
import torch.nn as nn

from . import _native as nat
from .ops import lora_linear

try:
    from safetensors.torch import safe_open
    from safetensors.torch import save_file as safe_save

    safetensors_available = True
except ImportError:  # reference: lora.py:12-29
    from .safe_open import safe_open

    def safe_save(tensors, filename, metadata=None):
        raise EnvironmentError(
            "Saving safetensors requires the safetensors library. Please install with pip or similar."
        )

    safetensors_available = False


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


# --------------------------------------------------------------------------------------------------
# module finders (enumeration order IS the on-disk index: lora.py:61-114)
# --------------------------------------------------------------------------------------------------
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
    params, names = [], []
    if loras is not None:
        loras = torch.load(loras, map_location="cpu", weights_only=True)
    for holder, name, child in _find_modules(model, target_replace_module, search_class=[nn.Linear]):
        wrapped = _wrap_linear(holder, name, child, r, follow_weight=False)
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


# --------------------------------------------------------------------------------------------------
# file formats (lora.py:201-407).  Host-side I/O; the byte layout is the parity artefact.
# --------------------------------------------------------------------------------------------------
def save_lora_weight(model, path="./lora.pt", target_replace_module=DEFAULT_TARGET_REPLACE):
    """`.pt`: flat python list [up0, down0, up1, ...] of fp16 CPU tensors (lora.py:201-213)."""
    flat = []
    for up, down in extract_lora_ups_down(model, target_replace_module=target_replace_module):
        flat += [up.weight.to("cpu").to(torch.float16), down.weight.to("cpu").to(torch.float16)]
    torch.save(flat, path)


def save_lora_as_json(model, path="./lora.json"):
    """Nested-list JSON dump of [up0, down0, ...] (lora.py:216-225)."""
    flat = []
    for up, down in extract_lora_ups_down(model):
        flat += [up.weight.detach().cpu().numpy().tolist(), down.weight.detach().cpu().numpy().tolist()]
    with open(path, "w") as f:
        json.dump(flat, f)


def save_safeloras_with_embeds(
    modelmap: Dict[str, Tuple[nn.Module, Set[str]]] = {},
    embeds: Dict[str, torch.Tensor] = {},
    outpath="./lora.safetensors",
):
    """One safetensors file for several models: tensors `{name}:{i}:up|down`, metadata `{name}` = JSON list of
    target classes and `{name}:{i}:rank`; TI embeddings stored under their token with metadata EMBED_FLAG
    (lora.py:228-258)."""
    tensors, meta = {}, {}
    for name, (model, targets) in modelmap.items():
        meta[name] = json.dumps(list(targets))
        for i, (up, down) in enumerate(extract_lora_ups_down(model, targets)):
            meta[f"{name}:{i}:rank"] = str(down.out_features)
            tensors[f"{name}:{i}:up"] = up.weight
            tensors[f"{name}:{i}:down"] = down.weight
    for token, tensor in (embeds or {}).items():  # save_all passes None when save_ti is off
        meta[token] = EMBED_FLAG
        tensors[token] = tensor
    print(f"Saving weights to {outpath}")
    safe_save(tensors, outpath, meta)


def save_safeloras(modelmap: Dict[str, Tuple[nn.Module, Set[str]]] = {}, outpath="./lora.safetensors"):
    return save_safeloras_with_embeds(modelmap=modelmap, outpath=outpath)


def convert_loras_to_safeloras_with_embeds(
    modelmap: Dict[str, Tuple[str, Set[str], int]] = {},
    embeds: Dict[str, torch.Tensor] = {},
    outpath="./lora.safetensors",
):
    """`.pt` lists → one safetensors file; modelmap values are (path, targets, rank) (lora.py:268-302)."""
    tensors, meta = {}, {}
    for name, (path, targets, rank) in modelmap.items():
        meta[name] = json.dumps(list(targets))
        for pos, weight in enumerate(torch.load(path, map_location="cpu", weights_only=True)):
            idx, kind = divmod(pos, 2)
            if kind == 0:
                meta[f"{name}:{idx}:rank"] = str(rank)
                tensors[f"{name}:{idx}:up"] = weight
            else:
                tensors[f"{name}:{idx}:down"] = weight
    for token, tensor in embeds.items():
        meta[token] = EMBED_FLAG
        tensors[token] = tensor
    print(f"Saving weights to {outpath}")
    safe_save(tensors, outpath, meta)


def convert_loras_to_safeloras(modelmap: Dict[str, Tuple[str, Set[str], int]] = {}, outpath="./lora.safetensors"):
    convert_loras_to_safeloras_with_embeds(modelmap=modelmap, outpath=outpath)


def parse_safeloras(safeloras) -> Dict[str, Tuple[List[nn.parameter.Parameter], List[int], List[str]]]:
    """Opened safetensors → {name: ([up0, down0, ...] Parameters, ranks, target classes)}; embeddings are
    skipped; a tensor group without metadata raises ValueError (lora.py:313-371)."""
    meta = safeloras.metadata()
    owner = lambda key: key.split(":")[0]
    keys = sorted(safeloras.keys(), key=owner)
    out = {}
    for name, group in groupby(keys, owner):
        info = meta.get(name)
        if not info:
            raise ValueError(f"Tensor {name} has no metadata - is this a Lora safetensor?")
        if info == EMBED_FLAG:
            continue
        group = list(group)
        ranks = [4] * (len(group) // 2)
        weights = [None] * len(group)
        for key in group:
            _, idx, direction = key.split(":")
            idx = int(idx)
            ranks[idx] = int(meta[f"{name}:{idx}:rank"])
            weights[2 * idx + (direction == "down")] = nn.parameter.Parameter(safeloras.get_tensor(key))
        out[name] = (weights, ranks, json.loads(info))
    return out


def parse_safeloras_embeds(safeloras) -> Dict[str, torch.Tensor]:
    """{token: tensor} for every entry flagged EMBED_FLAG (lora.py:374-392)."""
    meta = safeloras.metadata()
    return {k: safeloras.get_tensor(k) for k in safeloras.keys() if meta.get(k) == EMBED_FLAG}


def load_safeloras(path, device="cpu"):
    return parse_safeloras(safe_open(path, framework="pt", device=device))


def load_safeloras_embeds(path, device="cpu"):
    return parse_safeloras_embeds(safe_open(path, framework="pt", device=device))


def load_safeloras_both(path, device="cpu"):
    handle = safe_open(path, framework="pt", device=device)
    return parse_safeloras(handle), parse_safeloras_embeds(handle)


# --------------------------------------------------------------------------------------------------
# merge and patching (lora.py:410-600)
# --------------------------------------------------------------------------------------------------
def weight_apply_lora(model, loras, target_replace_module=DEFAULT_TARGET_REPLACE, alpha=1.0):
    """W ← W + α·(up @ down).type(W.dtype) for every target nn.Linear, as a NEW Parameter (lora.py:410-424).
    Runs as one HIP kernel per layer (csrc/optim.hip: merge_kernel); the weights must live on the device."""
    for _, _, child in _find_modules(model, target_replace_module, search_class=[nn.Linear]):
        weight = child.weight
        up = loras.pop(0).detach().to(weight.device)
        down = loras.pop(0).detach().to(weight.device)
        if not weight.is_cuda:
            raise RuntimeError(
                "weight_apply_lora: the merge runs as a HIP kernel; move the model to 'cuda' first "
                "(there is no CPU fallback)."
            )
        merged = weight.detach().clone().contiguous()
        held = up.dtype if up.dtype in (torch.float16, torch.bfloat16) else torch.float32
        nat.lora_merge_weight(merged, down.float().contiguous(), up.float().contiguous(), alpha, held)
        child.weight = nn.Parameter(merged)


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


def monkeypatch_or_replace_safeloras(models, safeloras):
    """Patch every model named in an opened safetensors file onto `models.<name>` (lora.py:541-551)."""
    for name, (lora, ranks, target) in parse_safeloras(safeloras).items():
        model = getattr(models, name, None)
        if not model:
            print(f"No model provided for {name}, contained in Lora")
            continue
        monkeypatch_or_replace_lora(model, lora, target, ranks)


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


def _derived_path(path: str, tag: str) -> str:
    assert path.endswith(".pt"), "Only .pt files are supported"
    return ".".join(path.split(".")[:-1] + [tag, "pt"])


def _text_lora_path(path: str) -> str:  # lora.py:603-605
    return _derived_path(path, "text_encoder")


def _ti_lora_path(path: str) -> str:  # lora.py:608-610
    return _derived_path(path, "ti")


# --------------------------------------------------------------------------------------------------
# textual-inversion embeddings and pipeline patching (lora.py:613-732) — host-side convenience
# --------------------------------------------------------------------------------------------------
def apply_learned_embed_in_clip(
    learned_embeds,
    text_encoder,
    tokenizer,
    token: Optional[Union[str, List[str]]] = None,
    idempotent=False,
):
    """Adds learned tokens to the tokenizer and writes their rows into the CLIP embedding table
    (lora.py:613-656)."""
    if isinstance(token, str):
        tokens = [token]
    elif isinstance(token, list):
        assert len(learned_embeds.keys()) == len(
            token
        ), "The number of tokens and the number of embeds should be the same"
        tokens = token
    else:
        tokens = list(learned_embeds.keys())

    for token in tokens:
        print(token)
        vector = learned_embeds[token]
        added = tokenizer.add_tokens(token)
        if not idempotent:
            suffix = 1
            while added == 0:
                print(f"The tokenizer already contains the token {token}.")
                token = f"{token[:-1]}-{suffix}>"
                print(f"Attempting to add the token {token}.")
                added = tokenizer.add_tokens(token)
                suffix += 1
        elif added == 0:
            print(f"The tokenizer already contains the token {token}.")
            print(f"Replacing {token} embedding.")
        text_encoder.resize_token_embeddings(len(tokenizer))
        row = tokenizer.convert_tokens_to_ids(token)
        text_encoder.get_input_embeddings().weight.data[row] = vector
    return token


def load_learned_embed_in_clip(
    learned_embeds_path,
    text_encoder,
    tokenizer,
    token: Optional[Union[str, List[str]]] = None,
    idempotent=False,
):
    learned = torch.load(learned_embeds_path, map_location="cpu", weights_only=True)
    apply_learned_embed_in_clip(learned, text_encoder, tokenizer, token, idempotent)


def patch_pipe(
    pipe,
    maybe_unet_path,
    token: Optional[str] = None,
    r: int = 4,
    patch_unet=True,
    patch_text=False,
    patch_ti=False,
    idempotent_token=True,
    unet_target_replace_module=DEFAULT_TARGET_REPLACE,
    text_target_replace_module=TEXT_ENCODER_DEFAULT_TARGET_REPLACE,
):
    """Loads a `.pt` triple or one `.safetensors` into a diffusers pipeline (lora.py:672-732)."""
    if maybe_unet_path.endswith(".pt"):
        if maybe_unet_path.endswith(".ti.pt"):
            unet_path = maybe_unet_path[:-6] + ".pt"
        elif maybe_unet_path.endswith(".text_encoder.pt"):
            unet_path = maybe_unet_path[:-16] + ".pt"
        else:
            unet_path = maybe_unet_path  # the reference leaves this case undefined (NameError)

        if patch_unet:
            print("LoRA : Patching Unet")
            monkeypatch_or_replace_lora(
                pipe.unet,
                torch.load(unet_path, map_location="cpu", weights_only=True),
                r=r,
                target_replace_module=unet_target_replace_module,
            )
        if patch_text:
            print("LoRA : Patching text encoder")
            monkeypatch_or_replace_lora(
                pipe.text_encoder,
                torch.load(_text_lora_path(unet_path), map_location="cpu", weights_only=True),
                target_replace_module=text_target_replace_module,
                r=r,
            )
        if patch_ti:
            print("LoRA : Patching token input")
            token = load_learned_embed_in_clip(
                _ti_lora_path(unet_path), pipe.text_encoder, pipe.tokenizer, token=token, idempotent=idempotent_token
            )
    elif maybe_unet_path.endswith(".safetensors"):
        handle = safe_open(maybe_unet_path, framework="pt", device="cpu")
        monkeypatch_or_replace_safeloras(pipe, handle)
        apply_learned_embed_in_clip(
            parse_safeloras_embeds(handle), pipe.text_encoder, pipe.tokenizer, token=token, idempotent=idempotent_token
        )


@torch.no_grad()
def inspect_lora(model):
    """{module name: [mean |up @ down|]} for every LoraInjectedLinear (lora.py:735-752)."""
    moved = {}
    for name, module in model.named_modules():
        if type(module).__name__ == "LoraInjectedLinear":
            delta = module.lora_up.weight.data @ module.lora_down.weight.data
            moved.setdefault(name, []).append(delta.flatten().abs().mean().item())
    return moved


def save_all(
    unet,
    text_encoder,
    placeholder_token_ids,
    placeholder_tokens,
    save_path,
    save_lora=True,
    save_ti=True,
    target_replace_module_text=TEXT_ENCODER_DEFAULT_TARGET_REPLACE,
    target_replace_module_unet=DEFAULT_TARGET_REPLACE,
    safe_form=True,
):
    """Saves LoRA (+TI embeddings) as a `.pt` triple or one `.safetensors` (lora.py:755-821)."""

    def learned_rows():
        rows = {}
        for tok, tok_id in zip(placeholder_tokens, placeholder_token_ids):
            vec = text_encoder.get_input_embeddings().weight[tok_id]
            print(f"Current Learned Embeddings for {tok}:, id {tok_id} ", vec[:4])
            rows[tok] = vec.detach().cpu()
        return rows

    if not safe_form:
        if save_ti:
            ti_path = _ti_lora_path(save_path)
            torch.save(learned_rows(), ti_path)
            print("Ti saved to ", ti_path)
        if save_lora:
            save_lora_weight(unet, save_path, target_replace_module=target_replace_module_unet)
            print("Unet saved to ", save_path)
            save_lora_weight(text_encoder, _text_lora_path(save_path), target_replace_module=target_replace_module_text)
            print("Text Encoder saved to ", _text_lora_path(save_path))
        return

    assert save_path.endswith(".safetensors"), f"Save path : {save_path} should end with .safetensors"
    loras, embeds = {}, None
    if save_lora:
        loras["unet"] = (unet, target_replace_module_unet)
        loras["text_encoder"] = (text_encoder, target_replace_module_text)
    if save_ti:
        embeds = learned_rows()
    save_safeloras_with_embeds(loras, embeds, save_path)
