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
