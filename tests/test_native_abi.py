"""The C-ABI library loads on a machine without a GPU and exports exactly what include/lora_hip.h declares."""
import ctypes
import os
import re

from diffusion_finetuning_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "lora_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:const\s+char\s*\*|int64_t|int)\s+(\w+)\s*\(", text, flags=re.M)))


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_functions()
    assert len(declared) >= 16 and "lora_linear_fwd" in declared and "ddpm_mse_fwd_bwd" in declared
    handle = ctypes.CDLL(nat.library_path())
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in lora_hip.h but not exported"
    assert sorted(nat.SIGNATURES) == declared  # the Python binding covers the whole header, nothing more


def test_version_and_status_strings_without_gpu():
    lib = nat.lib()
    assert lib.lora_version() == nat.ABI_VERSION == 9
    assert lib.lora_status_string(0) == b"ok"
    assert b"rank" in lib.lora_status_string(-2).lower()
    assert lib.lora_mse_workspace_bytes() >= 16 and lib.lora_sqnorm_workspace_bytes() >= 16


def test_argument_validation_needs_no_gpu():
    lib = nat.lib()
    # null pointers / bad rank are rejected before any HIP call
    assert lib.lora_linear_fwd(None, None, None, None, None, None, None, None, None, 4, 8, 8,2, 1.0, 1, None) == -1
    assert lib.lora_linear_fwd(None, None, None, None, None, None, None, None, None, 4, 8, 8,9, 1.0, 1, None) == -2
    assert lib.lora_linear_fwd(None, None, None, None, None, None, None, None, None, 4, 8, 8,2, 1.0, 7, None) == -1
    assert lib.lora_linear_bwd_params(None, None, None, None, None, None, 0, 1, 4, 8, 8, 0, 1.0, 1, None) == -2
    assert lib.lora_reduce_partials(None, 0, 1, None, 4, 0, None) == -1


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "diffusion_finetuning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_no_mfma_result_is_read_early_across_a_branch(tmp_path):
    """hipcc pads the distance between an MFMA and the first vector instruction that reads its destination registers along
    the LAYOUT order of the blocks only: a conditional branch that skips the padded block can land on a reader that comes
    too early — what made round 5's persistent-tile build of the fused GEMM return rows 16–31 / columns 2–3 of a tile wrong,
    differently on every run (tools/check_mfma_hazard.py).  Every kernel source is compiled to ISA and scanned."""
    import subprocess
    import sys
    from concurrent.futures import ThreadPoolExecutor

    from diffusion_finetuning_amd import build_native as bn

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_mfma_hazard as chk

    srcs = [s for s in bn.SOURCES if s not in ("prof.hip", "optim.hip", "ddpm_loss.hip", "embed.hip")]  # (kernels with MFMAs)

    def to_isa(src):
        out = str(tmp_path / (src + ".s"))
        subprocess.run([bn.HIPCC, *bn.FLAGS, *bn.SOURCE_FLAGS.get(src, []), "-S", "--cuda-device-only",
                        os.path.join(bn.CSRC, src), "-o", out], check=True, capture_output=True)
        return out

    with ThreadPoolExecutor(max_workers=4) as ex:
        files = list(ex.map(to_isa, srcs))
    findings = []
    for f in files:
        kernels = chk.parse(f)
        assert kernels, f
        findings += chk.check(kernels)
    assert not findings, findings[:3]
    # the checker itself: the hazard of the round-5 build, reduced to its shape, is found
    bad = tmp_path / "bad.s"
    bad.write_text("_Z3badv:\n\tv_mfma_f32_16x16x32_f16 v[2:5], v[40:43], v[36:39], v[16:19]\n\ts_cbranch_vccz .LBB0_2\n"
                   "\ts_nop 6\n\tv_pk_add_f32 v[2:3], v[8:9], v[2:3]\n.LBB0_2:\n\ts_add_i32 s18, s9, s63\n"
                   "\tv_cvt_pk_f16_f32 v5, v4, v5\n\ts_endpgm\n.Lfunc_end0:\n")
    assert len(chk.check(chk.parse(str(bad)))) == 1
