"""Pipeline-level conveniences of the `lora_diffusion` API (reference: lora_diffusion/lora.py:541-551, 613-821):
patching a diffusers pipeline from files, textual-inversion embeddings, inspection and `save_all`.  Host-side.
"""
from typing import List, Optional, Union

import torch

from .core import (
    DEFAULT_TARGET_REPLACE,
    TEXT_ENCODER_DEFAULT_TARGET_REPLACE,
    monkeypatch_or_replace_lora,
)
from .formats import (
    _text_lora_path,
    _ti_lora_path,
    parse_safeloras,
    parse_safeloras_embeds,
    safe_open,
    save_lora_weight,
    save_safeloras_with_embeds,
)


def monkeypatch_or_replace_safeloras(models, safeloras):
    """Patch every model named in an opened safetensors file onto `models.<name>` (lora.py:541-551)."""
    for name, (lora, ranks, target) in parse_safeloras(safeloras).items():
        model = getattr(models, name, None)
        if not model:
            print(f"No model provided for {name}, contained in Lora")
            continue
        monkeypatch_or_replace_lora(model, lora, target, ranks)


def _token_list(learned_embeds, token) -> List[str]:
    """Which names the learned vectors are registered under: the file's own keys, one override, or a list of them."""
    if token is None:
        return list(learned_embeds.keys())
    if isinstance(token, str):
        return [token]
    assert len(learned_embeds.keys()) == len(token), "The number of tokens and the number of embeds should be the same"
    return list(token)


def _register_token(tokenizer, token: str, idempotent: bool) -> str:
    """Adds `token` to the tokenizer.  If it exists already: keep it (idempotent — its row is overwritten) or find a
    free variant `<name-1>`, `<name-1-2>`, ... (the reference's renaming rule, lora.py:634-646).  Returns the name used."""
    if tokenizer.add_tokens(token) or idempotent:
        return token
    name, n = token, 1
    while True:  # "<s>" → "<s-1>" → "<s-1-2>" → ...: each retry extends the previous candidate, like the reference
        name = f"{name[:-1]}-{n}>"
        if tokenizer.add_tokens(name):
            break
        n += 1
    print(f"token {token} is taken; registered the embedding as {name}")
    return name


def apply_learned_embed_in_clip(
    learned_embeds,
    text_encoder,
    tokenizer,
    token: Optional[Union[str, List[str]]] = None,
    idempotent=False,
):
    """Registers learned textual-inversion vectors: each name goes into the tokenizer and its vector into the matching
    row of the CLIP input-embedding table (reference behaviour: lora.py:613-656).  Vectors are looked up under the
    REQUESTED name (so an override must be a key of `learned_embeds`, as in the reference); returns the last name
    actually registered."""
    used = None
    for name in _token_list(learned_embeds, token):
        vector = learned_embeds[name]
        used = _register_token(tokenizer, name, idempotent)
        text_encoder.resize_token_embeddings(len(tokenizer))
        table = text_encoder.get_input_embeddings().weight
        table.data[tokenizer.convert_tokens_to_ids(used)] = vector
    return used


def load_learned_embed_in_clip(
    learned_embeds_path,
    text_encoder,
    tokenizer,
    token: Optional[Union[str, List[str]]] = None,
    idempotent=False,
):
    learned = torch.load(learned_embeds_path, map_location="cpu", weights_only=True)
    apply_learned_embed_in_clip(learned, text_encoder, tokenizer, token, idempotent)


def _unet_pt_path(path: str) -> str:
    """`x.ti.pt` / `x.text_encoder.pt` / `x.pt` all name the same triple; its UNet member is `x.pt`."""
    for suffix in (".ti.pt", ".text_encoder.pt"):
        if path.endswith(suffix):
            return path[: -len(suffix)] + ".pt"
    return path  # (the reference leaves the plain `x.pt` case undefined: NameError)


def _load_list(path):
    return torch.load(path, map_location="cpu", weights_only=True)


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
    """Installs LoRA factors (and learned tokens) from files into a diffusers-style pipeline object with `.unet`,
    `.text_encoder`, `.tokenizer` (reference: lora.py:672-732).  A `.safetensors` file carries everything and is
    applied whole; a `.pt` path names a triple (`x.pt`, `x.text_encoder.pt`, `x.ti.pt`) of which the `patch_*` flags
    pick the members.  Other suffixes are ignored, as in the reference."""
    if maybe_unet_path.endswith(".safetensors"):
        handle = safe_open(maybe_unet_path, framework="pt", device="cpu")
        monkeypatch_or_replace_safeloras(pipe, handle)
        apply_learned_embed_in_clip(parse_safeloras_embeds(handle), pipe.text_encoder, pipe.tokenizer, token=token,
                                    idempotent=idempotent_token)
        return
    if not maybe_unet_path.endswith(".pt"):
        return
    unet_path = _unet_pt_path(maybe_unet_path)
    members = (
        (patch_unet, "unet", pipe.unet, unet_path, unet_target_replace_module),
        (patch_text, "text encoder", pipe.text_encoder, _text_lora_path(unet_path), text_target_replace_module),
    )
    for wanted, label, model, path, targets in members:
        if wanted:
            print(f"LoRA: patching the {label} from {path}")
            monkeypatch_or_replace_lora(model, _load_list(path), target_replace_module=targets, r=r)
    if patch_ti:
        print(f"LoRA: adding learned tokens from {_ti_lora_path(unet_path)}")
        load_learned_embed_in_clip(_ti_lora_path(unet_path), pipe.text_encoder, pipe.tokenizer, token=token,
                                   idempotent=idempotent_token)


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
