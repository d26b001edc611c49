"""Builds liblora_hip.so (gfx950) in-tree with hipcc.  `python -m diffusion_finetuning_amd.build_native`.

The library is plain C-ABI (include/lora_hip.h); it links only the HIP runtime, not torch.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "liblora_hip.so")
SOURCES = ["lora_gemm.hip", "lora_grad.hip", "ddpm_loss.hip", "optim.hip", "sandwich.hip", "attn_ctx.hip", "attn_flash.hip", "embed.hip", "prof.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-source flags (none at present).  A kernel that runs at ONE wave per SIMD (> 256 registers) wants
# ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]: there hipcc otherwise gives every MFMA an accumulation-register destination and copies each
# result out with v_accvgpr_read before the vector unit may touch it (profiles/r06_attn_dkdv_one_wave_and_32x32_kernels_rejected.hip).
SOURCE_FLAGS = {}


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "lora_hip.h")]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    for name in sorted(SOURCE_FLAGS):
        h.update((name + " " + " ".join(SOURCE_FLAGS[name])).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, variant: str = "", defines=()) -> str:
    """Builds the library; `variant` (dev tool, tools/build_variant.py) builds liblora_hip_<variant>.so with extra -D flags
    next to the real one for same-box A/B runs (DFA_LIB_PATH selects it)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    lib_path = os.path.join(LIB_DIR, f"liblora_hip_{variant}.so") if variant else LIB_PATH
    obj_dir = os.path.join(LIB_DIR, f"obj_{variant}") if variant else LIB_DIR
    os.makedirs(obj_dir, exist_ok=True)
    stamp = os.path.join(LIB_DIR, f"liblora_hip_{variant}.stamp" if variant else "liblora_hip.stamp")
    digest = _digest() + (" " + " ".join(defines) if defines else "")
    if not force and os.path.exists(lib_path) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return lib_path

    def compile_one(src: str) -> str:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, *SOURCE_FLAGS.get(src, []), *defines, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(digest)
    return lib_path


if __name__ == "__main__":
    # python -m diffusion_finetuning_amd.build_native [--force] [--variant NAME -DFOO=1 ...]
    args = [a for a in sys.argv[1:] if a != "--force"]
    name = ""
    if "--variant" in args:
        i = args.index("--variant")
        name = args[i + 1]
        del args[i:i + 2]
    print(build(force="--force" in sys.argv, variant=name, defines=tuple(args)))
