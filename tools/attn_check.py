"""Dev tool: the short-context attention kernels (csrc/attn_ctx.hip) against fp32 torch math, and timed against
the stock SDPA path the harness used before (split heads → padded SDPA → merge heads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffusion_finetuning_amd import _native as nat
from diffusion_finetuning_amd.sandwich import ctx_attention, split_heads, merge_heads
from torch.nn.attention import sdpa_kernel, SDPBackend

dev = "cuda"
torch.manual_seed(0)

def ref(q, k, v, H):
    B, Tq, HD = q.shape; Tk = k.shape[1]; d = HD // H
    qf, kf, vf = (t.float().view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qf @ kf.transpose(-1, -2) * d ** -0.5, dim=-1)
    return (p @ vf).transpose(1, 2).reshape(B, Tq, HD)

def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()

def check(B, Tq, Tk, H, d, dtype):
    q = torch.randn(B, Tq, H * d, device=dev).to(dtype).requires_grad_()
    k = torch.randn(B, Tk, H * d, device=dev).to(dtype).requires_grad_()
    v = torch.randn(B, Tk, H * d, device=dev).to(dtype).requires_grad_()
    g = torch.randn(B, Tq, H * d, device=dev).to(dtype)
    o = ctx_attention(q, k, v, H)
    dq, dk, dv = torch.autograd.grad(o, (q, k, v), g)
    qr, kr, vr = (t.detach().float().requires_grad_() for t in (q, k, v))
    orf = ref(qr, kr, vr, H)
    dqr, dkr, dvr = torch.autograd.grad(orf, (qr, kr, vr), g.float())
    errs = {"o": rel(o, orf), "dq": rel(dq, dqr), "dk": rel(dk, dkr), "dv": rel(dv, dvr)}
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    ok = all(e < tol for e in errs.values())
    print(("OK  " if ok else "FAIL"), B, Tq, Tk, H, d, str(dtype)[6:], {k_: f"{e:.1e}" for k_, e in errs.items()}, flush=True)
    return ok

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3

def timing(B, Tq, Tk, H, d):
    dtype = torch.float16
    q = torch.randn(B, Tq, H * d, device=dev).to(dtype).requires_grad_()
    k = torch.randn(B, Tk, H * d, device=dev).to(dtype).requires_grad_()
    v = torch.randn(B, Tk, H * d, device=dev).to(dtype).requires_grad_()
    g = torch.randn(B, Tq, H * d, device=dev).to(dtype)
    D = 64 if d < 64 else (128 if d < 128 else d)
    def stock():
        with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION]):
            o = F.scaled_dot_product_attention(split_heads(q, H, D), split_heads(k, H, D), split_heads(v, H, D), scale=d ** -0.5)
        return merge_heads(o, d)
    def mine():
        return ctx_attention(q, k, v, H)
    for name, f in (("stock", stock), ("hip", mine)):
        tf = bench(lambda: f().detach())
        def fb():
            o = f(); torch.autograd.grad(o, (q, k, v), g)
        tfb = bench(fb)
        print(f"  {name:6s} B={B} Tq={Tq} Tk={Tk} H={H} d={d}: fwd {tf:7.1f} us   fwd+bwd {tfb:7.1f} us", flush=True)

ok = True
for dtype in (torch.float16, torch.bfloat16):
    for (B, Tq, Tk, H, d) in [(2, 64, 77, 2, 40), (1, 100, 77, 3, 40), (2, 256, 77, 8, 80), (1, 50, 5, 1, 8), (2, 130, 96, 2, 64),
                              (1, 77, 128, 2, 96), (1, 16, 1, 1, 16), (4, 1024, 77, 8, 80), (4, 4096, 77, 8, 40), (1, 333, 100, 4, 48),
                              (4, 256, 77, 8, 160), (2, 64, 77, 8, 160), (1, 70, 90, 2, 104)]:
        ok &= check(B, Tq, Tk, H, d, dtype)
print("ALL OK" if ok else "SOME FAILED")
if "--time" in sys.argv:
    for shp in [(4, 4096, 77, 8, 40), (4, 1024, 77, 8, 80), (4, 256, 77, 8, 160)]:
        timing(*shp)
