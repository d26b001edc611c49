#!/bin/bash
# Dev tool: build a variant of the library next to the real one for same-box A/B runs (DFA_LIB_PATH selects it).
# usage: tools/build_variant.sh <name> [extra hipcc flags...]   ->  diffusion_finetuning_amd/lib/liblora_hip_<name>.so
name=$1; shift
cd "$(dirname "$0")/../diffusion_finetuning_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -shared -o ../lib/liblora_hip_$name.so \
  lora_gemm.hip lora_grad.hip ddpm_loss.hip optim.hip sandwich.hip attn_ctx.hip attn_flash.hip embed.hip prof.hip && echo built $name
