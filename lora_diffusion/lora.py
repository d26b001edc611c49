"""`lora_diffusion.lora` — the module the reference's CLIs import from — backed by the MI355X-native path."""
import diffusion_finetuning_amd.lora as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
