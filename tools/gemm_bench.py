"""Dev tool: per-shape device time of the hot-path kernels (dispatch-attached events), optional forced tile.
usage: [LORA_FORCE_TILE=0|1|2] python tools/gemm_bench.py [--grad] [--ref]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_finetuning_amd import _native as nat
dev = "cuda"
SHAPES = [(16384,320,320),(16384,320,2560),(16384,1280,320),(4096,640,640),(4096,640,5120),(4096,2560,640),(1024,1280,1280),(1024,1280,10240),(1024,5120,1280),(256,1280,1280),(308,768,320),(308,768,1280)]
def run(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    nat.prof_enable(iters + 4)
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    res = nat.prof_collect(); nat.prof_enable(0)
    tot = sum(v["ms"] for v in res.values()); n = sum(v["launches"] for v in res.values())
    return 1e3 * tot / n, list(res.keys())
def main():
    dtype = torch.float16
    print("tile override:", os.environ.get("LORA_FORCE_TILE"))
    for (M,K,N) in SHAPES:
        x = torch.randn(M,K,device=dev).to(dtype); w = (torch.randn(N,K,device=dev)/K**0.5).to(dtype); wt = w.t().contiguous()
        a = torch.randn(4,K,device=dev)/4; b = torch.randn(N,4,device=dev)*0.05; dy = torch.randn(M,N,device=dev).to(dtype)
        y, t = nat.lora_linear_fwd(x,w,None,a,b,1.0); dx,u = nat.lora_linear_bwd_input(dy,wt,a,b,1.0,True)
        ga = torch.zeros(4,K,device=dev); gb = torch.zeros(N,4,device=dev)
        tf, kf = run(lambda: nat.lora_linear_fwd(x,w,None,a,b,1.0))
        tb, kb = run(lambda: nat.lora_linear_bwd_input(dy,wt,a,b,1.0,True))
        tg, kg = run(lambda: nat.lora_linear_bwd_params(dy,x,t,u,ga,gb,1.0))
        fl = 2.0*M*K*N; by = 2.0*(M*K+N*K+M*N); bg = 2.0*(M*N+M*K)
        line = f"{M:6d}x{K:5d}x{N:6d} fwd {tf:7.1f}us {fl/tf/1e6:6.0f}TF {by/tf/1e3:6.0f}GB/s [{kf[0][18:28]}] | bwd {tb:7.1f}us {fl/tb/1e6:6.0f}TF [{kb[0][18:28]}] | grad {tg:6.1f}us {bg/tg/1e3:6.0f}GB/s"
        if "--ref" in sys.argv:
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            for _ in range(5): torch.nn.functional.linear(x,w)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): torch.nn.functional.linear(x,w)
            e1.record(); torch.cuda.synchronize()
            line += f" | hipblaslt(host-timed) {e0.elapsed_time(e1)/50*1e3:6.1f}us"
        print(line, flush=True)
main()
