"""LoRA file formats of the `lora_diffusion` API (reference: lora_diffusion/lora.py:12-29, 201-407, 603-610).

Host-side I/O only, byte-compatible with the reference: `.pt` = flat python list [up0, down0, ...] of fp16 CPU
tensors; safetensors = tensors `{name}:{i}:up|down` + metadata `{name}` (JSON list of target classes) and
`{name}:{i}:rank`, textual-inversion embeddings stored under their token with metadata EMBED_FLAG.
"""
import json
from itertools import groupby
from typing import Dict, List, Set, Tuple

import torch
import torch.nn as nn

from .core import DEFAULT_TARGET_REPLACE, EMBED_FLAG, extract_lora_ups_down

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


def lerp_lora_lists(l1: List[torch.Tensor], l2: List[torch.Tensor], alpha: float = 0.5) -> List[torch.Tensor]:
    """LoRA ⊕ LoRA of `lora_add --mode lpl` (lora_diffusion/cli_lora_add.py:44-60) on two positional
    `[up0, down0, up1, ...]` lists: every tensor of `l1` becomes alpha·l1 + (1-alpha)·l2 IN PLACE (`.data` is
    replaced, dtype and device kept, op-by-op rounding as the reference) and the list of merged tensors is returned,
    ready for `torch.save`.  One HIP launch per dtype over the concatenated lists (host tensors are staged through
    the device).  A trailing unpaired tensor is dropped, like the reference's pair zip."""
    from . import _native as nat

    n = 2 * min(len(l1) // 2, len(l2) // 2)
    if n == 0:
        return []
    for x1, x2 in zip(l1[:n], l2[:n]):
        if x1.shape != x2.shape:
            raise RuntimeError(f"lerp_lora_lists: LoRA tensors differ in shape: {tuple(x1.shape)} vs {tuple(x2.shape)}")
    device = nat.staging_device(*l1[:n])
    for dtype in {t.dtype for t in l1[:n]}:
        idx = [i for i in range(n) if l1[i].dtype == dtype]
        a = torch.cat([l1[i].detach().to(device).reshape(-1) for i in idx])
        b = torch.cat([l2[i].detach().to(device, dtype).reshape(-1) for i in idx])
        nat.lora_lerp_(a, b, alpha)
        off = 0
        for i in idx:
            k = l1[i].numel()
            l1[i].data = a[off:off + k].view(l1[i].shape).to(l1[i].device, copy=True)
            off += k
    return list(l1[:n])


def _derived_path(path: str, tag: str) -> str:
    assert path.endswith(".pt"), "Only .pt files are supported"
    return ".".join(path.split(".")[:-1] + [tag, "pt"])


def _text_lora_path(path: str) -> str:  # lora.py:603-605
    return _derived_path(path, "text_encoder")


def _ti_lora_path(path: str) -> str:  # lora.py:608-610
    return _derived_path(path, "ti")
