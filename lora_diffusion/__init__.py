"""Drop-in package name.  `from lora_diffusion import inject_trainable_lora, ...` keeps working for the
reference's trainers; everything resolves to diffusion_finetuning_amd."""
from diffusion_finetuning_amd import *  # noqa: F401,F403
from diffusion_finetuning_amd.lora import _find_children, _find_modules, _text_lora_path, _ti_lora_path  # noqa: F401
