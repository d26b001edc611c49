"""Dev experiment: SDPA fwd+bwd time for the SD1.5 self-attention shapes under different backends / head-dim padding."""
import torch, time, sys
import torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
dev="cuda"
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for (N,d) in [(4096,40),(1024,80),(256,160)]:
    for pad in (0, 64 if d<64 else (128 if d<128 else 192)):
        D = pad or d
        q=torch.randn(4,8,N,D,device=dev,dtype=torch.float16,requires_grad=True); k=torch.randn_like(q,requires_grad=True); v=torch.randn_like(q,requires_grad=True)
        if pad:
            with torch.no_grad():
                q[...,d:]=0; k[...,d:]=0; v[...,d:]=0
        scale = d ** -0.5
        for name, be in (("flash",SDPBackend.FLASH_ATTENTION),("efficient",SDPBackend.EFFICIENT_ATTENTION),("math",SDPBackend.MATH)):
            try:
                with sdpa_kernel(be):
                    def f():
                        o=F.scaled_dot_product_attention(q,k,v,scale=scale); o.backward(o)
                    def ffwd():
                        with torch.no_grad(): F.scaled_dot_product_attention(q,k,v,scale=scale)
                    t=bench(f); tf=bench(ffwd)
                print(f"N={N} d={d} D={D} {name}: fwd {tf*1e3:.0f}us fwd+bwd {t*1e3:.0f}us", flush=True)
            except Exception as e:
                print(f"N={N} d={d} D={D} {name}: FAILED {str(e)[:80]}", flush=True)
