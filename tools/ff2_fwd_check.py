"""Dev tool: the frozen second linear layer of a FeedForward block (ff.net.2 forward, bias included) as stock F.linear against the
library's fused GEMM with zero LoRA factors (bias in the epilogue), host-timed over 50 launches, cold-ish (one big read between)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_finetuning_amd import _native as nat
dev = "cuda"; dt = torch.float16
flush = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device=dev)
def timed(fn, n=30):
    for _ in range(3): fn()
    tot = 0.0
    for _ in range(n):
        flush.view(torch.float32).sum()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    return tot / n * 1e3
for (M, F, N) in [(16384, 1280, 320), (4096, 2560, 640), (1024, 5120, 1280), (256, 5120, 1280)]:
    x = torch.randn(M, F, device=dev).to(dt); w = (torch.randn(N, F, device=dev) / F ** 0.5).to(dt); b = torch.randn(N, device=dev).to(dt)
    z = torch.zeros(16 * max(F, N), dtype=dt, device=dev)
    out = torch.empty(M, N, dtype=dt, device=dev)
    t_stock = timed(lambda: torch.nn.functional.linear(x, w, b))
    t_lib = timed(lambda: nat.lora_gemm_packed(x, F, w, b, z, z, None, None, 0, out, None, M, F, N, 1, 0.0))
    ref = torch.nn.functional.linear(x, w, b)
    err = ((out.float() - ref.float()).norm() / ref.float().norm()).item()
    print(f"ff.net.2 fwd {M}x{F}->{N}: F.linear {t_stock:6.1f} us | library GEMM (zero factors, fused bias) {t_lib:6.1f} us | rel diff {err:.1e}", flush=True)
