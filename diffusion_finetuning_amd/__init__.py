"""MI355X-native LoRA fine-tuning hot path with the `lora_diffusion` API surface."""
