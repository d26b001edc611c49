"""SURVEY §8 f-4, second half: the GEGLU gate inside the forward kernel's epilogue (`lora_linear_geglu_fwd`).

Reference arithmetic: diffusers `GEGLU.forward` — `hidden, gate = proj(x).chunk(2, -1); hidden * gelu(gate)` — around the
`proj` LoraInjectedLinear (target class "GEGLU", lora_diffusion/lora.py:53; operator lora.py:49-50)."""
import pytest
import torch

import diffusion_finetuning_amd as dfa
from diffusion_finetuning_amd import _native as nat

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(M, K, F, r, bias, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(2 * F, K, generator=g) / K ** 0.5
    b = torch.randn(2 * F, generator=g) * 0.3 if bias else None
    a = torch.randn(r, K, generator=g) / r
    up = torch.randn(2 * F, r, generator=g) * 0.05
    dev = lambda t, dt=dtype: None if t is None else t.to(DEV).to(dt)
    return dev(x), dev(w), dev(b), dev(a, torch.float32), dev(up, torch.float32)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,F,r,bias", [
    (1024, 320, 1280, 4, True),     # SD1.5 level-0 proj at reduced M
    (300, 640, 2560, 4, True),      # ragged rows
    (256, 1280, 5120, 16, True),    # mid block, rank 16
    (129, 64, 64, 1, False),        # one K-step, one column tile, no bias
    (4096, 320, 1280, 8, True),
])
def test_gated_forward_equals_the_two_launches_it_replaces_and_float64(close, dtype, M, K, F, r, bias):
    x, w, b, a, up = _case(M, K, F, r, bias, dtype)
    packs = nat.lora_pack_factors(a, up, dtype)
    y_ref, t_ref = nat.lora_linear_fwd(x, w, b, a, up, 0.7, packs)
    out_ref = nat.geglu_gate_fwd(y_ref)
    res = nat.lora_linear_geglu_fwd(x, w, b, r, 0.7, packs, True)
    assert res is not None
    out, y, t = res
    # same contraction order per element, same rounding of y, same gate function: bit for bit
    assert torch.equal(y, y_ref) and torch.equal(t, t_ref) and torch.equal(out, out_ref)
    # without a backward pass y is never written
    out2, y2, t2 = nat.lora_linear_geglu_fwd(x, w, b, r, 0.7, packs, False)
    assert y2 is None and torch.equal(out2, out) and torch.equal(t2, t)
    # and against float64 math on the same (rounded) operands
    xd, wd = x.double().cpu(), w.double().cpu()
    ad, bd = a.to(dtype).double().cpu(), up.to(dtype).double().cpu()
    yd = xd @ wd.t() + (b.double().cpu() if b is not None else 0.0) + 0.7 * (xd @ ad.t()) @ bd.t()
    tol = 2e-3 if dtype == torch.float16 else 1.2e-2
    close(y, yd, tol, "y")
    h, gt = y.double().cpu().chunk(2, dim=-1)
    close(out, h * torch.nn.functional.gelu(gt), tol, "out")  # the gate itself: from the stored y, one rounding
    close(t, xd @ a.double().cpu().to(dtype).double().t(), tol, "T")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_gate_function_over_the_whole_range(dtype):
    """16-bit tensors evaluate Φ(g) with a one-rcp-one-exp erfc form (csrc/common.h): over g ∈ [−12, 12] the gate and its
    backward stay within one ulp of the storage type of float64 exact-erf gelu (absolute floor 4e-7 in the far negative tail,
    where gelu itself is below 1e-5)."""
    g = torch.linspace(-12, 12, 48001).to(dtype)
    g = g[: g.numel() // 8 * 8]
    h = torch.full_like(g, 1.5)
    y = torch.cat([h.view(-1, 8), g.view(-1, 8)], dim=1).to(DEV)
    out = nat.geglu_gate_fwd(y).cpu().double().view(-1)
    gd, hd = g.double(), h.double()
    ref = hd * torch.nn.functional.gelu(gd)
    ulp = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7, torch.float32: 2.0 ** -22}[dtype]
    assert ((out - ref).abs() <= ref.abs() * ulp + 4e-7).all()
    d = torch.ones_like(y[:, :8])
    dy = nat.geglu_gate_bwd(y, d).cpu().double()
    Phi = 0.5 * (1 + torch.erf(gd / 2 ** 0.5))
    dgelu = Phi + gd * torch.exp(-0.5 * gd * gd) / (2 * torch.pi) ** 0.5
    assert ((dy[:, :8].reshape(-1) - gd * Phi).abs() <= (gd * Phi).abs() * ulp + 4e-7).all()
    assert ((dy[:, 8:].reshape(-1) - hd * dgelu).abs() <= (hd * dgelu).abs() * ulp + 1e-6).all()


def test_shapes_without_a_fused_kernel_are_reported_not_faked():
    for (M, K, F, dtype) in [(64, 64, 96, torch.float16), (64, 72, 64, torch.float16), (64, 64, 64, torch.float32)]:
        x, w, b, a, up = _case(M, K, F, 4, True, dtype)
        packs = nat.lora_pack_factors(a, up, dtype)
        assert nat.lora_linear_geglu_fwd(x, w, b, 4, 1.0, packs, True) is None


class _GEGLU(torch.nn.Module):  # the caller: class name and layout of diffusers' GEGLU
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = torch.nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * torch.nn.functional.gelu(gate)


_GEGLU.__name__ = "GEGLU"


@pytest.mark.parametrize("mode", ["half_model", "autocast_fp16", "fp32"])
def test_geglu_hook_routes_proj_through_the_gated_kernel_and_backpropagates(close, relerr, monkeypatch, mode):
    """`set_use_hip_geglu` on a GEGLU whose `proj` is a LoraInjectedLinear: forward is ONE launch, backward the gate's
    streaming kernel + the LoRA backward; checked against float64 autograd of the reference's composite.  fp32 has no
    fused kernel and must take the two-launch route with the same results."""
    from diffusion_finetuning_amd.attention import set_use_hip_geglu

    torch.manual_seed(5)
    K, F, M = 320, 1280, 777
    mod = _GEGLU(K, F)
    mod.requires_grad_(False)
    params, _ = dfa.inject_trainable_lora(mod, target_replace_module={"GEGLU"}, r=4)
    with torch.no_grad():
        mod.proj.lora_up.weight.normal_(0, 0.05)
    x = torch.randn(2, M, K)
    dout = torch.randn(2, M, F)
    # float64 reference: the reference's operator + GEGLU composite
    lin, dn, upm = mod.proj.linear, mod.proj.lora_down, mod.proj.lora_up
    cd = torch.float32 if mode == "fp32" else torch.float16
    xr = x.to(cd).double().requires_grad_(True)
    A = dn.weight.detach().double().requires_grad_(True)
    Bm = upm.weight.detach().double().requires_grad_(True)
    Wq = lin.weight.detach().to(cd).double()
    bq = lin.bias.detach().to(cd).double()
    yr = xr @ Wq.t() + bq + (xr @ A.t()) @ Bm.t()
    hr, gr = yr.chunk(2, dim=-1)
    (hr * torch.nn.functional.gelu(gr)).backward(dout.to(cd).double())

    mod = mod.to(DEV)
    if mode == "half_model":
        mod = mod.half()
    assert set_use_hip_geglu(mod) == 1
    xg = x.to(DEV).to(torch.float16 if mode == "half_model" else torch.float32).requires_grad_(True)
    calls = {"gate": 0, "fused": 0}
    real_gate, real_fused = nat.geglu_gate_fwd, nat.lora_linear_geglu_fwd

    def count(name, fn):
        def wrapped(*a, **k):
            res = fn(*a, **k)
            calls[name] += res is not None
            return res
        return wrapped

    monkeypatch.setattr(nat, "geglu_gate_fwd", count("gate", real_gate))
    monkeypatch.setattr(nat, "lora_linear_geglu_fwd", count("fused", real_fused))
    if mode == "autocast_fp16":
        with torch.autocast("cuda", dtype=torch.float16):
            out = mod(xg)
    else:
        out = mod(xg)
    out.backward(dout.to(DEV).to(out.dtype))
    # 16-bit: ONE forward launch (GEMM + LoRA + gate); fp32: the two launches it stands for
    assert calls == ({"gate": 1, "fused": 0} if mode == "fp32" else {"gate": 0, "fused": 1}), calls
    tol = 1e-5 if mode == "fp32" else 3e-3
    close(out, (hr * torch.nn.functional.gelu(gr)).detach(), tol, "out")
    close(xg.grad, xr.grad, tol, "dx")
    assert relerr(mod.proj.lora_down.weight.grad, A.grad) < tol
    assert relerr(mod.proj.lora_up.weight.grad, Bm.grad) < tol
    # no-grad forward: same values, nothing saved
    with torch.no_grad():
        if mode == "autocast_fp16":
            with torch.autocast("cuda", dtype=torch.float16):
                out_ng = mod(xg)
        else:
            out_ng = mod(xg)
    assert torch.equal(out_ng, out)
    # the switch comes off again
    assert set_use_hip_geglu(mod, False) == 1
