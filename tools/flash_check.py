"""Dev tool: the long-context attention kernels (csrc/attn_flash.hip) against fp32 torch math, timed against stock SDPA."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
from diffusion_finetuning_amd import _native as nat
from diffusion_finetuning_amd.sandwich import split_heads, merge_heads

dev = "cuda"
lib = ctypes.CDLL(nat.library_path())
vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
lib.attn_flash_fwd.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, cf, ci, vp]
lib.attn_flash_bwd.argtypes = [vp] * 10 + [ci, ci, ci, ci, ci, cf, ci, vp]
DT = {torch.float16: 1, torch.bfloat16: 2}

def flash_fwd(q, k, v, H):
    B, Tq, HD = q.shape; Tk = k.shape[1]; d = HD // H
    o = torch.empty_like(q); lse = torch.empty(B, H, Tq, device=dev, dtype=torch.float32)
    rc = lib.attn_flash_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Tq, Tk, H, d,
                            d ** -0.5, DT[q.dtype], torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return o, lse

def flash_bwd(q, k, v, o, do, lse, H):
    B, Tq, HD = q.shape; Tk = k.shape[1]; d = HD // H
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(B * H * Tq, device=dev, dtype=torch.float32)
    rc = lib.attn_flash_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), B, Tq, Tk, H, d, d ** -0.5,
                            DT[q.dtype], torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return dq, dk, dv

def ref(q, k, v, H):
    B, Tq, HD = q.shape; d = HD // H
    qf, kf, vf = (t.float().view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) * d ** -0.5
    lse = torch.logsumexp(s, dim=-1) * 1.4426950408889634
    return (torch.softmax(s, dim=-1) @ vf).transpose(1, 2).reshape(B, Tq, HD), lse

def rel(a, b): return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3

torch.manual_seed(0)
ok = True
for dtype in (torch.float16, torch.bfloat16):
    for (B, Tq, Tk, H, d) in [(1, 64, 64, 1, 40), (2, 200, 130, 2, 40), (1, 256, 256, 2, 64), (2, 1024, 1024, 4, 80),
                              (1, 256, 256, 8, 160), (1, 70, 300, 2, 96), (1, 100, 77, 2, 128), (2, 4096, 4096, 8, 40)]:
        q = torch.randn(B, Tq, H * d, device=dev).to(dtype); k = torch.randn(B, Tk, H * d, device=dev).to(dtype)
        v = torch.randn(B, Tk, H * d, device=dev).to(dtype)
        o, lse = flash_fwd(q, k, v, H)
        o_r, lse_r = ref(q, k, v, H)
        go = torch.randn(B, Tq, H * d, device=dev).to(dtype)
        dq, dk, dv = flash_bwd(q, k, v, o, go, lse, H)
        qr, kr, vr = (t.float().requires_grad_() for t in (q, k, v))
        o_rr, _ = ref(qr, kr, vr, H)
        dq_r, dk_r, dv_r = torch.autograd.grad(o_rr, (qr, kr, vr), go.float())
        e = (rel(o, o_r), rel(lse, lse_r), rel(dq, dq_r), rel(dk, dk_r), rel(dv, dv_r))
        tol = 4e-3 if dtype == torch.float16 else 3e-2
        good = e[0] < tol and e[1] < 1e-3 and max(e[2:]) < 2 * tol
        ok &= good
        print("OK  " if good else "FAIL", B, Tq, Tk, H, d, str(dtype)[6:], f"o {e[0]:.1e} lse {e[1]:.1e} dq {e[2]:.1e} dk {e[3]:.1e} dv {e[4]:.1e}", flush=True)
print("ALL OK" if ok else "SOME FAILED")
if "--time" in sys.argv:
    for (B, T, H, d) in [(4, 4096, 8, 40), (4, 1024, 8, 80), (4, 256, 8, 160)]:
        q = torch.randn(B, T, H * d, device=dev).half(); k = torch.randn_like(q); v = torch.randn_like(q)
        D = 64 if d < 64 else (128 if d < 128 else d)
        def stock():
            with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION]):
                o = F.scaled_dot_product_attention(split_heads(q, H, D), split_heads(k, H, D), split_heads(v, H, D), scale=d ** -0.5)
            return merge_heads(o, d)
        with torch.no_grad():
            print(f"T={T} d={d}: stock fwd {bench(stock):.0f} us   flash fwd {bench(lambda: flash_fwd(q, k, v, H)):.0f} us", flush=True)
        qs, ks_, vs = (t.clone().requires_grad_() for t in (q, k, v))
        go = torch.randn_like(q)
        def stock_fb():
            with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION]):
                o = F.scaled_dot_product_attention(split_heads(qs, H, D), split_heads(ks_, H, D), split_heads(vs, H, D), scale=d ** -0.5)
            torch.autograd.grad(merge_heads(o, d), (qs, ks_, vs), go)
        o, lse = flash_fwd(q, k, v, H)
        def mine_fb():
            o_, lse_ = flash_fwd(q, k, v, H); flash_bwd(q, k, v, o_, go, lse_, H)
        print(f"          stock fwd+bwd {bench(stock_fb):.0f} us   flash fwd+bwd {bench(mine_fb):.0f} us", flush=True)
