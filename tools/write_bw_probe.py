"""Dev probe: how fast can 126 MB be WRITTEN on this chip (the bytes the 2560-tile gated `proj` forward stores: y [16384, 2560] + gated
[16384, 1280], f16)?  torch fill_ / zero_ of the same byte count, and of one third / three times of it, timed with events over back-to-back
launches.  The fused GEMM's 128-row class takes 35.5 µs per launch with its main loop switched off (profiles/r06_gemm_skeleton_time_per_class.log)."""
import torch

dev = "cuda"
for mb in (42, 126, 378):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.float16, device=dev)
    for name, fn in (("zero_", lambda: x.zero_()), ("fill_(1)", lambda: x.fill_(1.0))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / reps
        print(f"{mb:4d} MB {name:9s} {us:7.1f} us  {mb / us * 1e-3 * 1e3:6.2f} TB/s" if False else f"{mb:4d} MB {name:9s} {us:7.1f} us  {mb * 1e6 / (us * 1e-6) / 1e12:5.2f} TB/s")
# a strided pattern like the GEMM epilogue's: 128-byte pieces of rows 5120 bytes apart (copy_ into a column slice)
y = torch.empty((16384, 2560), dtype=torch.float16, device=dev)
src = torch.ones((16384, 64), dtype=torch.float16, device=dev)
for _ in range(3):
    y[:, 0:64].copy_(src)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for c in range(0, 2560, 64):
    y[:, c:c + 64].copy_(src)
e.record()
torch.cuda.synchronize()
print(f"84 MB as 40 column-slice copies of 128-byte row pieces (one launch each): {1e3 * s.elapsed_time(e):.1f} us in total")
