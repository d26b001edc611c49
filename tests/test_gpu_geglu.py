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
    # grids where the 128×160 gated tile (80 h + 80 g columns, C tile in two passes) is the cheaper one: 640 tiles of 128
    # would need two rounds on the chip, 512 tiles of 160 one (csrc/lora_gemm.hip: gate_tile_width); with a ragged row count
    (1024, 128, 5120, 4, True),
    (1000, 64, 5120, 16, False),
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


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,Nz,F", [(1024, 320, 1280), (300, 640, 2560), (256, 1280, 5120), (129, 64, 128),
                                    (4096, 64, 2560), (4000, 128, 2560)])  # the last two: 128×160 tiles (two store passes)
def test_gate_backward_in_the_epilogue_of_the_following_linear_layers_backward(close, dtype, M, Nz, F):
    """`geglu_linear_bwd`: dY = gate-backward(dZ·W2, Y) in one launch.  Bit-identical to the library's own two steps (the same
    GEMM kernel without a gate, then geglu_gate_bwd on the rounded dout), and close to float64 autograd of
    z = (h·gelu(g)) @ W2ᵀ."""
    g = torch.Generator().manual_seed(M + F)
    y = torch.randn(M, 2 * F, generator=g).to(dtype)
    w2 = (torch.randn(Nz, F, generator=g) / F ** 0.5).to(dtype)
    dz = torch.randn(M, Nz, generator=g).to(dtype)
    yd, w2d, dzd = y.to(DEV), w2.to(DEV), dz.to(DEV)
    w2t = nat.lora_cast_matrix(w2d, dtype, True)  # [F, Nz]
    dy = nat.geglu_linear_bwd(dzd, w2t, yd)
    assert dy is not None and dy.shape == (M, 2 * F)
    # the two steps it stands for, on the same kernel: dout = dZ·W2 through lora_gemm_packed with zero factors, then the gate
    zf = torch.zeros(16 * max(Nz, F), dtype=dtype, device=DEV)
    dout = torch.empty(M, F, dtype=dtype, device=DEV)
    nat.lora_gemm_packed(dzd, Nz, w2t, None, zf, zf, None, None, 0, dout, None, M, Nz, F, 1, 0.0)
    assert torch.equal(dy, nat.geglu_gate_bwd(yd, dout))
    # float64 autograd of the reference composite
    yr = y.double().requires_grad_(True)
    h, gt = yr.chunk(2, dim=-1)
    ((h * torch.nn.functional.gelu(gt)) @ w2.double().t()).backward(dz.double())
    close(dy, yr.grad, 3e-3 if dtype == torch.float16 else 2e-2, "dY")


def test_gate_backward_shapes_without_a_fused_kernel_are_reported():
    for (M, Nz, F, dtype) in [(64, 64, 96, torch.float16), (64, 72, 128, torch.float16), (64, 64, 128, torch.float32)]:
        y = torch.randn(M, 2 * F, device=DEV).to(dtype)
        w2t = torch.randn(F, Nz, device=DEV).to(dtype)
        dz = torch.randn(M, Nz, device=DEV).to(dtype)
        assert nat.geglu_linear_bwd(dz, w2t, y) is None


class _FeedForward(torch.nn.Module):  # diffusers FeedForward layout: net = [GEGLU, Dropout, Linear]
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = torch.nn.ModuleList([_GEGLU(dim, dim * mult), torch.nn.Dropout(0.0), torch.nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


_FeedForward.__name__ = "FeedForward"


@pytest.mark.parametrize("mode", ["half_model", "autocast_fp16", "fp32"])
def test_feed_forward_hook_folds_both_halves_of_the_gate(close, relerr, monkeypatch, mode):
    """`set_use_hip_geglu` on a FeedForward(GEGLU) block: forward = the gated `proj` launch + the stock second linear layer,
    backward = `geglu_linear_bwd` (no gate kernel at all) + the LoRA backward; against float64 autograd of the reference
    composite.  fp32 has no fused kernels and takes the unfused steps with the same results; without gradients, or with active
    dropout, the module's own forward runs."""
    from diffusion_finetuning_amd.attention import set_use_hip_geglu

    torch.manual_seed(9)
    K, M = 320, 515
    ff = _FeedForward(K)
    ff.requires_grad_(False)
    dfa.inject_trainable_lora(ff, target_replace_module={"GEGLU"}, r=4)
    proj = ff.net[0].proj
    with torch.no_grad():
        proj.lora_up.weight.normal_(0, 0.05)
    x = torch.randn(2, M, K)
    dz = torch.randn(2, M, K)
    cd = torch.float32 if mode == "fp32" else torch.float16
    xr = x.to(cd).double().requires_grad_(True)
    A = proj.lora_down.weight.detach().double().requires_grad_(True)
    Bm = proj.lora_up.weight.detach().double().requires_grad_(True)
    W1, b1 = proj.linear.weight.detach().to(cd).double(), proj.linear.bias.detach().to(cd).double()
    W2, b2 = ff.net[2].weight.detach().to(cd).double(), ff.net[2].bias.detach().to(cd).double()
    yr = xr @ W1.t() + b1 + (xr @ A.t()) @ Bm.t()
    hr, gr = yr.chunk(2, dim=-1)
    zr = (hr * torch.nn.functional.gelu(gr)) @ W2.t() + b2
    zr.backward(dz.to(cd).double())

    ff = ff.to(DEV)
    if mode == "half_model":
        ff = ff.half()
    assert set_use_hip_geglu(ff) == 1 and "forward" in ff.__dict__
    calls = {"gate_bwd": 0, "fused_bwd": 0}
    real_gate, real_fused = nat.geglu_gate_bwd, nat.geglu_linear_bwd

    def count(name, fn):
        def wrapped(*a, **k):
            res = fn(*a, **k)
            calls[name] += res is not None
            return res
        return wrapped

    monkeypatch.setattr(nat, "geglu_gate_bwd", count("gate_bwd", real_gate))
    monkeypatch.setattr(nat, "geglu_linear_bwd", count("fused_bwd", real_fused))
    xg = x.to(DEV).to(torch.float16 if mode == "half_model" else torch.float32).requires_grad_(True)
    if mode == "autocast_fp16":
        with torch.autocast("cuda", dtype=torch.float16):
            z = ff(xg)
    else:
        z = ff(xg)
    z.backward(dz.to(DEV).to(z.dtype))
    assert calls == ({"gate_bwd": 1, "fused_bwd": 0} if mode == "fp32" else {"gate_bwd": 0, "fused_bwd": 1}), calls
    tol = 2e-5 if mode == "fp32" else 4e-3
    close(z, zr.detach(), tol, "z")
    close(xg.grad, xr.grad, tol, "dx")
    assert relerr(proj.lora_down.weight.grad, A.grad) < tol and relerr(proj.lora_up.weight.grad, Bm.grad) < tol
    with torch.no_grad():  # no gradient wanted: the module's own forward (its GEGLU keeps the forward-only fusion)
        if mode == "autocast_fp16":
            with torch.autocast("cuda", dtype=torch.float16):
                z_ng = ff(xg)
        else:
            z_ng = ff(xg)
    close(z_ng, zr.detach(), tol, "z no-grad")
    assert set_use_hip_geglu(ff, False) == 1 and "forward" not in ff.__dict__


@pytest.mark.parametrize("reentrant", [False, True])
def test_feed_forward_hook_under_activation_checkpointing(relerr, reentrant):
    """train_lora_dreambooth.py:627-630 (`--gradient_checkpointing`): the hooked block re-runs its forward inside backward;
    gradients must equal the un-checkpointed ones bit for bit (same kernels, same order)."""
    from torch.utils.checkpoint import checkpoint

    from diffusion_finetuning_amd.attention import set_use_hip_geglu

    torch.manual_seed(4)
    ff = _FeedForward(320)
    ff.requires_grad_(False)
    dfa.inject_trainable_lora(ff, target_replace_module={"GEGLU"}, r=4)
    with torch.no_grad():
        ff.net[0].proj.lora_up.weight.normal_(0, 0.05)
    ff = ff.to(DEV).half()
    set_use_hip_geglu(ff)
    x = torch.randn(2, 300, 320, device=DEV).half()
    dz = torch.randn(2, 300, 320, device=DEV).half()
    params = [ff.net[0].proj.lora_down.weight, ff.net[0].proj.lora_up.weight]

    def run(ckpt):
        for p in params:
            p.grad = None
        xi = x.clone().requires_grad_(True)
        z = checkpoint(ff, xi, use_reentrant=reentrant) if ckpt else ff(xi)
        z.backward(dz)
        return z.detach(), xi.grad, [p.grad.clone() for p in params]

    z0, dx0, g0 = run(False)
    z1, dx1, g1 = run(True)
    assert torch.equal(z0, z1) and torch.equal(dx0, dx1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
