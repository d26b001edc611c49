"""Dev tool: per-shape device time of the hot-path kernels (dispatch-attached events).
usage: [LORA_FORCE_TILE=0|2] [LORA_FORCE_STAGES=2|3|4|6] python tools/gemm_bench.py [--grouped] [--grads] [--ref]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_finetuning_amd import _native as nat
dev = "cuda"
SHAPES = [(16384,320,320),(16384,320,2560),(16384,1280,320),(4096,640,640),(4096,640,5120),(4096,2560,640),(1024,1280,1280),(1024,1280,10240),(1024,5120,1280),(256,1280,1280),(256,1280,10240),(308,768,320),(308,768,1280)]
COLD = "--cold" in sys.argv or "--cold-read" in sys.argv  # in-model conditions: weights come from HBM (a 600 MB write evicts L2 + Infinity Cache), the row operand was just written
_flush = None
PREFETCH = [None]
def run(fn, iters=20, warm=None):
    global _flush
    iters = int(os.environ.get("GB_ITERS", iters))
    if COLD:
        if _flush is None: _flush = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device=dev)
        inner = fn
        def fn():
            if "--cold-read" in sys.argv: _flush.view(torch.float32).sum()  # evict with CLEAN lines (no write-backs competing with the kernel)
            else: _flush.zero_()
            if warm is not None: warm.mul_(1.0)
            if "--prefetch" in sys.argv and PREFETCH[0] is not None:  # premise test: the weight read once (→ Infinity Cache) before the launch
                PREFETCH[0].view(torch.float32).view(-1)[::32].sum()
            inner()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    nat.prof_enable(16 * iters + 64)
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    res = nat.prof_collect(); nat.prof_enable(0)
    tot = sum(v["ms"] for v in res.values()); n = sum(v["launches"] for v in res.values())
    keys = list(res.keys())
    if "other" in res and len(res) > 1:  # split-K: show the reduction launch's share
        keys = [k for k in keys if k != "other"] + [f"+reduce {1e3 * res['other']['ms'] / iters:.1f}us"]
    return 1e3 * tot / iters, n / iters, keys
def per_shape():
    dtype = torch.float16
    print("tile", os.environ.get("LORA_FORCE_TILE"), "stages", os.environ.get("LORA_FORCE_STAGES"))
    sel = os.environ.get("GB_SHAPES")
    for (M,K,N) in ([SHAPES[int(i)] for i in sel.split(",")] if sel else SHAPES):
        x = torch.randn(M,K,device=dev).to(dtype); w = (torch.randn(N,K,device=dev)/K**0.5).to(dtype); wt = w.t().contiguous()
        a = torch.randn(4,K,device=dev)/4; b = torch.randn(N,4,device=dev)*0.05; dy = torch.randn(M,N,device=dev).to(dtype)
        y, t = nat.lora_linear_fwd(x,w,None,a,b,1.0); dx,u = nat.lora_linear_bwd_input(dy,wt,a,b,1.0,True)
        ga = torch.zeros(4,K,device=dev); gb = torch.zeros(N,4,device=dev)
        PREFETCH[0] = w
        tf, _, kf = run(lambda: nat.lora_linear_fwd(x,w,None,a,b,1.0), warm=x)
        PREFETCH[0] = wt
        tb, _, kb = run(lambda: nat.lora_linear_bwd_input(dy,wt,a,b,1.0,True), warm=dy)
        fl = 2.0*M*K*N; by = 2.0*(M*K+N*K+M*N)
        line = f"{M:6d}x{K:5d}x{N:6d} fwd {tf:7.1f}us {fl/tf/1e6:6.0f}TF {by/tf/1e3:6.0f}GB/s [{kf[0][18:28]}] | bwd {tb:7.1f}us {fl/tb/1e6:6.0f}TF [{kb[0][18:28]}] {kb[-1] if len(kb) > 1 else ''}"
        if "--ref" in sys.argv:
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            for _ in range(5): torch.nn.functional.linear(x,w)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): torch.nn.functional.linear(x,w)
            e1.record(); torch.cuda.synchronize()
            line += f" | hipblaslt(host-timed) {e0.elapsed_time(e1)/50*1e3:6.1f}us"
        print(line, flush=True)
def grouped():
    """q/k/v as one launch (block-diagonal rank 12) vs three; all 32 context K/V projections as one launch vs 32."""
    dtype = torch.float16
    for (M,K,N) in [(16384,320,320),(4096,640,640),(1024,1280,1280),(256,1280,1280)]:
        G, r = 3, 4
        x = torch.randn(M,K,device=dev).to(dtype); w = (torch.randn(G*N,K,device=dev)/K**0.5).to(dtype); wt = w.t().contiguous()
        dy = torch.randn(M,G*N,device=dev).to(dtype)
        Fa = torch.zeros(16,K,device=dev,dtype=dtype); Fa[:12] = torch.randn(12,K,device=dev).to(dtype)/4
        Qb = torch.zeros(G*N,16,device=dev,dtype=dtype); Fb = torch.zeros(16,G*N,device=dev,dtype=dtype); Qa = Fa.t().contiguous()
        for g in range(G):
            blk = (torch.randn(N,r,device=dev)*0.05).to(dtype); Qb[g*N:(g+1)*N, g*r:(g+1)*r] = blk; Fb[g*r:(g+1)*r, g*N:(g+1)*N] = blk.t()
        y = torch.empty(M,G*N,device=dev,dtype=dtype); t = torch.empty(M,12,device=dev); dx = torch.empty(M,K,device=dev,dtype=dtype); u = torch.empty(M,12,device=dev)
        tf, _, kf = run(lambda: nat.lora_gemm_packed(x,K,w,None,Fa,Qb,None,None,0,y,t,M,K,G*N,12,1.0), warm=x)
        tb, _, kb = run(lambda: nat.lora_gemm_packed(dy,G*N,wt,None,Fb,Qa,None,None,0,dx,u,M,G*N,K,12,1.0), warm=dy)
        by = 2.0*(M*K+G*N*K+M*G*N)
        print(f"qkv {M:6d}x{K:5d}x3*{N:5d} fwd {tf:7.1f}us {by/tf/1e3:6.0f}GB/s [{kf[0][18:28]}] | bwd {tb:7.1f}us {by/tb/1e3:6.0f}GB/s [{kb[0][18:28]}] {kb[-1] if len(kb) > 1 else ''}", flush=True)
    M, K, r = 308, 768, 4
    widths = [320]*10 + [640]*10 + [1280]*12
    total = sum(widths); G = len(widths)
    x = torch.randn(M,K,device=dev).to(dtype); w = (torch.randn(total,K,device=dev)/K**0.5).to(dtype)
    A16 = (torch.randn(G*16,K,device=dev)/4).to(dtype); B16 = (torch.randn(total,16,device=dev)*0.05).to(dtype); Bt16 = (torch.randn(16*total,device=dev)*0.05).to(dtype)
    tp, offs, o = [], [], 0
    for i, n in enumerate(widths):
        tp += [i | (1 << 16)] + [i]*(n//64-1); offs.append(o); o += n
    tile_part = torch.tensor(tp,dtype=torch.int32).to(dev)
    pt = torch.tensor([[offs[i], widths[i], 16*offs[i], i*M*r] for i in range(G)],dtype=torch.int64).to(dev)
    Y = torch.empty(M,total,device=dev,dtype=dtype); T = torch.empty(G,M,r,device=dev); U = torch.empty(G,M,r,device=dev); dY = torch.randn(M,total,device=dev).to(dtype)
    tf, _, kf = run(lambda: nat.lora_gemm_packed(x,K,w,None,A16,B16,tile_part,None,G,Y,T,M,K,total,r,1.0))
    tb, _, kb = run(lambda: nat.lora_gemm_packed(dY,total,None,None,Bt16,None,None,pt,G,None,U,M,64,0,r,1.0,work_cols=total))
    by = 2.0*(M*K+total*K+M*total)
    print(f"ctx-kv group {M}x{K}x{total} fwd {tf:7.1f}us {by/tf/1e3:6.0f}GB/s [{kf[0][18:28]}] | U-only bwd {tb:7.1f}us {2.0*M*total/tb/1e3:6.0f}GB/s [{kb[0][18:28]}]", flush=True)
def grads():
    """All 288 factor-gradient problems of one SD1.5 cfg-2 step in one lora_grad_batched call (+ the fold)."""
    dtype = torch.float16
    layers = [(16384,320,320)]*30 + [(16384,320,2560)]*5 + [(308,768,320)]*10 + [(4096,640,640)]*30 + [(4096,640,5120)]*5 + [(308,768,640)]*10 + [(1024,1280,1280)]*30 + [(1024,1280,10240)]*5 + [(308,768,1280)]*12 + [(256,1280,1280)]*6 + [(256,1280,10240)]*1
    r = int(os.environ.get("GB_RANK", 4))
    total = sum(r*(K+N) for _,K,N in layers); stride = (total+3)//4*4
    partials = torch.empty(nat.GRAD_MAX_BLOCKS, stride, device=dev); grads_ = torch.zeros(stride, device=dev)
    cache = {}
    def buf(M, C):
        if (M,C) not in cache: cache[(M,C)] = [torch.randn(M,C,device=dev).to(dtype) for _ in range(3)]
        return cache[(M,C)]
    probs, ranges, keep, off, nbytes = [], [], [], 0, 0
    for i,(M,K,N) in enumerate(layers):
        dy = buf(M,N)[i%3]; x = buf(M,K)[(i+1)%3]; t = torch.randn(M,r,device=dev); u = torch.randn(M,r,device=dev)
        keep += [t,u]
        only = os.environ.get("GB_GRADS_ONLY", "")  # "gB": the dYᵀ·T halves alone, "gA": the Uᵀ·X halves alone
        if only != "gA": probs.append(nat.grad_problem(dy,0,N,N,t,0,r,r,[partials.data_ptr()+4*off],r,False,stride,M,1.0))
        if only != "gB": probs.append(nat.grad_problem(x,0,K,K,u,0,r,r,[partials.data_ptr()+4*(off+N*r)],r,True,stride,M,1.0))
        ranges.append([off, r*(K+N), nat.grad_row_blocks(M), 0]); off += r*(K+N); nbytes += 2.0*M*(K+N)
    table = torch.tensor(ranges,dtype=torch.int64).to(dev)
    d0 = torch.device(dev,0)
    tg, nl, kg = run(lambda: nat.lora_grad_batched(probs, dtype, d0), iters=10)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    nat.lora_fold_partials(table,len(ranges),max(r_[1] for r_ in ranges),partials,stride,grads_,True); torch.cuda.synchronize(); e0.record()
    for _ in range(20): nat.lora_fold_partials(table,len(ranges),max(r_[1] for r_ in ranges),partials,stride,grads_,True)
    e1.record(); torch.cuda.synchronize()
    print(f"batched grads: {len(probs)} problems, {nl:.0f} launches, {tg:8.1f} us per step ({nbytes/tg/1e3:6.0f} GB/s algorithmic) | fold {e0.elapsed_time(e1)/20*1e3:6.1f} us (host-timed)", flush=True)
def geglu():
    """`proj` forward with the gate in the epilogue vs GEMM + gate kernel (host-timed pairs, graph-free)."""
    dtype = torch.float16
    for (M,K,F) in [(16384,320,1280),(4096,640,2560),(1024,1280,5120),(256,1280,5120)]:
        x = torch.randn(M,K,device=dev).to(dtype); w = (torch.randn(2*F,K,device=dev)/K**0.5).to(dtype); b = torch.randn(2*F,device=dev).to(dtype)
        a = torch.randn(4,K,device=dev)/4; up = torch.randn(2*F,4,device=dev)*0.05
        packs = nat.lora_pack_factors(a, up, dtype)
        def timed(fn, n=50):
            for _ in range(5): fn()
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
        def two():
            y,_ = nat.lora_linear_fwd(x,w,b,a,up,1.0,packs); nat.geglu_gate_fwd(y)
        t_gemm, _, _ = run(lambda: nat.lora_linear_fwd(x,w,b,a,up,1.0,packs))
        t_f, _, _ = run(lambda: nat.lora_linear_geglu_fwd(x,w,b,4,1.0,packs,True))
        t_fn, _, _ = run(lambda: nat.lora_linear_geglu_fwd(x,w,b,4,1.0,packs,False))
        print(f"proj {M:6d}x{K:5d}x2*{F:5d}: GEMM {t_gemm:6.1f}us | gated (y kept) {t_f:6.1f}us | gated (no y) {t_fn:6.1f}us || host-timed: two launches {timed(two):6.1f}us, fused {timed(lambda: nat.lora_linear_geglu_fwd(x,w,b,4,1.0,packs,True)):6.1f}us", flush=True)
        # backward: dY = gate-bwd(dz·W2, y) fused against stock mm + the gate kernel (host-timed pairs)
        w2 = (torch.randn(K,F,device=dev)/F**0.5).to(dtype); w2t = w2.t().contiguous(); dz = torch.randn(M,K,device=dev).to(dtype)
        y,_ = nat.lora_linear_fwd(x,w,b,a,up,1.0,packs)
        t_fb, _, _ = run(lambda: nat.geglu_linear_bwd(dz, w2t, y))
        def two_b():
            nat.geglu_gate_bwd(y, dz @ w2)
        print(f"      backward: fused {t_fb:6.1f}us (kernel) || host-timed: stock mm + gate {timed(two_b):6.1f}us, fused {timed(lambda: nat.geglu_linear_bwd(dz, w2t, y)):6.1f}us", flush=True)
if "--grouped" in sys.argv: grouped()
elif "--geglu" in sys.argv: geglu()
elif "--grads" in sys.argv: grads()
else: per_shape()
