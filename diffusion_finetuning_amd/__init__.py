"""MI355X-native LoRA fine-tuning hot path with the `lora_diffusion` API surface.

`from diffusion_finetuning_amd import *` gives every public name of the reference's lora_diffusion/lora.py.
"""
from .lora import *  # noqa: F401,F403
from .lora import (  # noqa: F401  (underscore names the reference's callers import explicitly)
    _find_children,
    _find_modules,
    _find_modules_old,
    _find_modules_v2,
    _text_lora_path,
    _ti_lora_path,
)
from .ops import ddpm_mse_loss, invalidate_weight_cache, lora_linear  # noqa: F401
