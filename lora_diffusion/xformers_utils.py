"""`lora_diffusion.xformers_utils` — the module the reference's trainers import their attention switch from
(training_scripts/train_lora_dreambooth.py:40,623-625) — backed by the HIP short-context attention core."""
from diffusion_finetuning_amd.attention import (  # noqa: F401
    set_use_hip_attention,
    set_use_hip_geglu,
    set_use_memory_efficient_attention_xformers,
    test_xformers_backwards,
)
