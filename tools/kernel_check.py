"""Dev tool: checks the HIP kernels against fp64 torch math on the GPU box and times them."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_finetuning_amd import _native as nat

dev = "cuda"
torch.manual_seed(0)

def ref(x, w, bias, a, b, s, dy):
    xd, wd, ad, bd, dyd = (t.double() for t in (x, w, a, b, dy))
    t = xd @ ad.T
    y = xd @ wd.T + s * (t @ bd.T)
    if bias is not None:
        y = y + bias.double()
    u = dyd @ bd
    dx = dyd @ wd + s * (u @ ad)
    gb = s * dyd.T @ t
    ga = s * u.T @ xd
    return y, t, u, dx, ga, gb

def rel(a, b):
    return ((a.double() - b).norm() / (b.norm() + 1e-30)).item()

def check(M, K, N, r, dtype, bias, s=0.7):
    x = torch.randn(M, K, device=dev)
    w = (torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5
    a = torch.randn(r, K, device=dev) / r
    b = torch.randn(N, r, device=dev) * 0.05
    bi = torch.randn(N, device=dev) * 0.1 if bias else None
    dy = torch.randn(M, N, device=dev)
    xc, wc, dyc = x.to(dtype), w.to(dtype), dy.to(dtype)
    bic = bi.to(dtype) if bias else None
    # reference sees the rounded operands (A,B rounded like the kernel does for 16-bit)
    ar = a.to(dtype).float() if dtype != torch.float32 else a
    br = b.to(dtype).float() if dtype != torch.float32 else b
    y_r, t_r, u_r, dx_r, ga_r, gb_r = ref(xc, wc, bic, ar, br, s, dyc)
    y, t = nat.lora_linear_fwd(xc, wc, bic, a, b, s)
    wt = nat.lora_cast_matrix(wc, dtype, True)
    assert torch.equal(wt, wc.t().contiguous())
    dx, u = nat.lora_linear_bwd_input(dyc, wt, a, b, s, True)
    _, u2 = nat.lora_linear_bwd_input(dyc, None, a, b, s, False)
    ga = torch.zeros(r, K, device=dev); gb = torch.zeros(N, r, device=dev)
    nat.lora_linear_bwd_params(dyc, xc, t, u, ga, gb, s)
    torch.cuda.synchronize()
    errs = dict(y=rel(y, y_r), t=rel(t, t_r), u=rel(u, u_r), u2=rel(u2, u_r), dx=rel(dx, dx_r), ga=rel(ga, ga_r), gb=rel(gb, gb_r))
    tol = 2e-5 if dtype == torch.float32 else (3e-3 if dtype == torch.float16 else 2e-2)
    ok = all(v < tol for v in errs.values())
    print(("OK  " if ok else "FAIL"), M, K, N, r, str(dtype).split(".")[1], bias, {k: f"{v:.1e}" for k, v in errs.items()}, flush=True)
    return ok

def bench(M, K, N, r, dtype, iters=50):
    x = torch.randn(M, K, device=dev).to(dtype); w = torch.randn(N, K, device=dev).to(dtype) / K ** 0.5
    a = torch.randn(r, K, device=dev) / r; b = torch.randn(N, r, device=dev) * 0.05
    dy = torch.randn(M, N, device=dev).to(dtype); wt = w.t().contiguous()
    y, t = nat.lora_linear_fwd(x, w, None, a, b, 1.0)
    dx, u = nat.lora_linear_bwd_input(dy, wt, a, b, 1.0, True)
    ga = torch.zeros(r, K, device=dev); gb = torch.zeros(N, r, device=dev)
    e = 2 if dtype != torch.float32 else 4
    res = {}
    def timeit(fn):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3  # us
    tf = timeit(lambda: nat.lora_linear_fwd(x, w, None, a, b, 1.0))
    tb = timeit(lambda: nat.lora_linear_bwd_input(dy, wt, a, b, 1.0, True))
    tg = timeit(lambda: nat.lora_linear_bwd_params(dy, x, t, u, ga, gb, 1.0))
    tt = timeit(lambda: torch.nn.functional.linear(x, w))
    bytes_f = e * (M * K + N * K + M * N); fl = 2.0 * M * K * N
    bytes_g = e * (M * N + M * K)
    print(f"bench {M}x{K}x{N} r{r} {str(dtype).split('.')[1]}: fwd {tf:.1f}us ({bytes_f/tf/1e6:.2f} TB/s, {fl/tf/1e6:.0f} TF/s) | bwd_in {tb:.1f}us ({fl/tb/1e6:.0f} TF/s) | grads {tg:.1f}us ({bytes_g/tg/1e6:.2f} TB/s) | torch F.linear {tt:.1f}us", flush=True)

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    allok = True
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        for (M, K, N, r, bias) in [(64, 32, 32, 4, False), (77, 64, 96, 4, True), (200, 320, 320, 4, True), (308, 768, 320, 1, False),
                                   (256, 1280, 1280, 8, True), (1024, 320, 2560, 16, True), (130, 40, 24, 3, True), (64, 36, 20, 2, False),
                                   (4096, 640, 640, 4, False), (1, 32, 32, 4, True)]:
            allok &= check(M, K, N, r, dtype, bias)
    print("ALL OK" if allok else "SOME FAILED")
    if "--bench" in sys.argv:
        for shp in [(16384, 320, 320), (16384, 320, 2560), (16384, 1280, 320), (4096, 640, 640), (4096, 640, 5120), (1024, 1280, 1280), (1024, 1280, 10240), (308, 768, 320), (256, 1280, 1280)]:
            bench(*shp, 4, torch.float16)
    sys.exit(0 if allok else 1)
