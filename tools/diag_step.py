"""Dev tool: where does a train step spend its (first-call) time?  Prints progress continuously."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
def log(*a):
    print(f"[{time.perf_counter()-T0:8.2f}s]", *a, flush=True)
T0 = time.perf_counter()
dev = "cuda"
dt = torch.float16
log("torch", torch.__version__, torch.cuda.get_device_name(0))
def timed(name, fn, n=3):
    t = time.perf_counter(); fn(); torch.cuda.synchronize(); first = time.perf_counter() - t
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); rest = (time.perf_counter() - t) / n
    log(f"{name}: first {first*1e3:.1f} ms, steady {rest*1e3:.2f} ms")
# convs (B=4)
for (cin, cout, hw, k, s) in [(4,320,64,3,1),(320,320,64,3,1),(320,320,64,3,2),(320,640,32,3,1),(640,640,32,3,1),(640,1280,16,3,1),(1280,1280,16,3,1),(1280,1280,8,3,1),(2560,1280,8,3,1),(2560,1280,16,3,1),(1920,1280,16,3,1),(1920,640,32,3,1),(1280,640,32,3,1),(960,640,32,3,1),(960,320,64,3,1),(640,320,64,3,1),(320,320,64,1,1)]:
    x = torch.randn(4, cin, hw, hw, device=dev, dtype=dt, requires_grad=True)
    w = torch.randn(cout, cin, k, k, device=dev, dtype=dt)
    def f():
        y = F.conv2d(x, w, padding=k//2, stride=s); y.backward(y)
    timed(f"conv {cin}->{cout} {hw}x{hw} k{k} s{s} fwd+bwd_data", f)
# attention
for (n, h, d, m) in [(4096,8,40,4096),(4096,8,40,77),(1024,8,80,1024),(1024,8,80,77),(256,8,160,256),(64,8,160,64)]:
    q = torch.randn(4,h,n,d, device=dev, dtype=dt, requires_grad=True); k_ = torch.randn(4,h,m,d, device=dev, dtype=dt, requires_grad=True); v = torch.randn(4,h,m,d, device=dev, dtype=dt, requires_grad=True)
    def f():
        o = F.scaled_dot_product_attention(q,k_,v); o.backward(o)
    timed(f"sdpa n{n} m{m} d{d}", f)
# groupnorm
for (c, hw) in [(320,64),(640,32),(1280,16),(2560,8)]:
    x = torch.randn(4,c,hw,hw, device=dev, dtype=dt, requires_grad=True); g = torch.nn.GroupNorm(32,c).to(dev).to(dt)
    def f():
        y = F.silu(g(x)); y.backward(y)
    timed(f"groupnorm+silu c{c} {hw}", f)
log("building model")
import bench
unet = bench.build_model(torch.device("cuda",0), dt, 4)
log("model built")
from diffusion_finetuning_amd.trainer import LoraTrainer
tr = LoraTrainer(unet, lr=1e-4)
data = bench.synthetic_steps(6, 4, 64, 0, 1, torch.device("cuda",0))
for i in range(6):
    t = time.perf_counter(); l = tr.step(*data[i]); torch.cuda.synchronize(); log(f"step {i}: {1e3*(time.perf_counter()-t):.1f} ms loss {l.item():.4f}")
log("fwd only timing")
x = torch.randn(4,4,64,64, device=dev, dtype=dt); ts = torch.randint(0,1000,(4,),device=dev); ctx = torch.randn(4,77,768,device=dev,dtype=dt)
with torch.no_grad():
    timed("unet fwd no_grad", lambda: unet(x, ts, ctx))
