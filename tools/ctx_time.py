"""Dev tool: device time of the short-context attention kernels alone (back-to-back launches, events), cold-ish operands.
usage: [CTX_BWD_WGS=n CTX_FWD_WGS=n] python tools/ctx_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_finetuning_amd import _native as nat

dev = "cuda"
torch.manual_seed(0)
flush = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device=dev)

def timed(fn, n=30):
    for _ in range(3): fn()
    tot = 0.0
    for _ in range(n):
        flush.fill_(1)  # evict L2 / Infinity Cache
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3

for (B, Tq, Tk, H, d) in [(4, 4096, 77, 8, 40), (4, 1024, 77, 8, 80), (4, 256, 77, 8, 160), (4, 64, 77, 8, 160)]:
    q = torch.randn(B, Tq, H * d, device=dev).half(); k = torch.randn(B, Tk, H * d, device=dev).half()
    v = torch.randn(B, Tk, H * d, device=dev).half(); g = torch.randn(B, Tq, H * d, device=dev).half()
    s = d ** -0.5
    tf = timed(lambda: nat.attn_ctx_fwd(q, k, v, H, s))
    tb = timed(lambda: nat.attn_ctx_bwd(q, k, v, g, H, s))
    print(f"B={B} Tq={Tq} Tk={Tk} H={H} d={d}: fwd {tf:6.1f} us   bwd (+reduce, + 4 allocations) {tb:6.1f} us", flush=True)
