"""Dev experiment: does the CK flash-attention library in this PyTorch build beat AOTriton on the SD shapes?"""
import torch, sys
import torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
dev = "cuda"
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for lib in ("aotriton", "ck"):
    try:
        torch.backends.cuda.preferred_rocm_fa_library(lib)
        print("library:", torch.backends.cuda.preferred_rocm_fa_library(), flush=True)
    except Exception as e:
        print("cannot select", lib, str(e)[:200], flush=True); continue
    for (N, Nk, d, D) in [(4096, 4096, 40, 64), (4096, 77, 40, 64), (1024, 1024, 80, 128), (256, 256, 160, 160)]:
        q = torch.randn(4, 8, N, D, device=dev, dtype=torch.float16, requires_grad=True)
        k = torch.randn(4, 8, Nk, D, device=dev, dtype=torch.float16, requires_grad=True); v = torch.randn_like(k, requires_grad=True)
        for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)):
            try:
                with sdpa_kernel(be):
                    def f():
                        o = F.scaled_dot_product_attention(q, k, v, scale=d ** -0.5); o.backward(o)
                    def ffwd():
                        with torch.no_grad(): F.scaled_dot_product_attention(q, k, v, scale=d ** -0.5)
                    t = bench(f); tf = bench(ffwd)
                print(f"  N={N} Nk={Nk} D={D} {name}: fwd {tf*1e3:.0f}us fwd+bwd {t*1e3:.0f}us", flush=True)
            except Exception as e:
                print(f"  N={N} Nk={Nk} D={D} {name}: FAILED {str(e)[:120]}", flush=True)
