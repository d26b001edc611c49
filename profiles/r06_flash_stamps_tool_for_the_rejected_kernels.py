"""Dev tool: shader-clock stamps of one tile of one wave of the narrow dK/dV kernel (a -DFLASH_STAMP=1 build of the library,
DFA_LIB_PATH selects it).  Prints the in-kernel clock (s_memtime vs the 100-MHz s_memrealtime) and the cycles per phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_finetuning_amd import _native as nat
torch.manual_seed(0)
B, T, H, d = 4, 4096, 8, 40
q = torch.randn(B, T, H * d, device="cuda").half(); k = torch.randn_like(q); v = torch.randn_like(q); go = torch.randn_like(q)
o, lse = nat.attn_flash_fwd(q, k, v, H, d ** -0.5)
for _ in range(50): nat.attn_flash_bwd(q, k, v, o, go, lse, H, d ** -0.5)
torch.cuda.synchronize()
lib = ctypes.CDLL(nat.library_path())
buf = (ctypes.c_ulonglong * 64)()
assert lib.lora_flash_read_stamps(buf) == 0
s = list(buf)
cyc, real = s[2] - s[0], s[3] - s[1]
print(f"kernel life of the stamped wave: {cyc} shader cycles, {real * 10} ns -> {cyc / (real * 10):.2f} GHz", flush=True)
names = ["tile top", "p0 A done", "p0 B done", "p0 rows+stores issued", "p0 C head", "p0 done", "barrier passed", "fetch issued",
         "-", "p1 A done", "p1 B done", "p1 rows issued", "p1 C head", "p1 done"]
st = s[4:20]
prev = st[0]
for i, n in enumerate(names):
    if n == "-": continue
    t = st[i]
    print(f"  {n:28s} +{(t - prev) & 0xffffffff:6d}   (since tile top {(t - st[0]) & 0xffffffff:6d})")
    prev = t
