"""Dev tool: torch.profiler view of one train step (which aten ops own the GPU time outside the hot path).
--shapes: group by input shapes; --stack OP: print the Python call sites of one aten op (e.g. aten::copy_);
--kernel PATTERN: attribute every GPU kernel whose name contains PATTERN (e.g. elementwise_kernel_manual_unroll) to the aten op
that launched it, its input shapes and the first call site inside this repository (forward ops; backward ops show the autograd
node's name instead of a site)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from diffusion_finetuning_amd.trainer import LoraTrainer
dev = torch.device("cuda", 0)
unet = bench.build_model(dev, torch.float16, 4)
tr = LoraTrainer(unet, lr=1e-4)
data = bench.synthetic_steps(4, 4, 64, 0, 1, dev)
for i in range(3): tr.step(*data[i])
torch.cuda.synchronize()
stack_op = sys.argv[sys.argv.index("--stack") + 1] if "--stack" in sys.argv else None
kern_pat = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else None
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
             with_stack=stack_op is not None or kern_pat is not None) as prof:
    tr.step(*data[3]); torch.cuda.synchronize()
def _site(ev):
    for fr in (ev.stack or []):
        for mark in ("diffusion_finetuning_amd/", "harness/"):
            if mark in fr and "torch_profile_step" not in fr:
                return fr[fr.index(mark):][:80]
    return "(autograd / no repo frame)"
if kern_pat:
    import collections
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        ks = [k for k in getattr(ev, "kernels", []) if kern_pat in k.name]
        if not ks: continue
        # innermost op only: an op whose child also owns these kernels would double count
        if any(any(kern_pat in k.name for k in getattr(ch, "kernels", [])) for ch in (ev.cpu_children or [])): continue
        key = (ev.name, str(ev.input_shapes)[:60], _site(ev))
        agg[key][0] += len(ks); agg[key][1] += sum(k.duration for k in ks)
    tot_n = sum(v[0] for v in agg.values()); tot_t = sum(v[1] for v in agg.values())
    print(f"kernels matching {kern_pat!r}: {tot_n} launches, {tot_t:.1f} us in this step")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{t:9.1f} us  x{n:3d}  {k[0]:28s} {k[1]:60s} {k[2]}")
    sys.exit(0)
if stack_op:
    import collections
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name == stack_op and ev.device_time_total > 0:
            site = next((s for s in ev.stack if "/repo/" in s and "torch_profile_step" not in s), (ev.stack or ["?"])[0])
            key = (site.split("/repo/")[-1][:90], str(ev.input_shapes)[:70])
            agg[key][0] += 1; agg[key][1] += ev.device_time_total
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{t:9.1f} us  x{n:3d}  {k[0]}  {k[1]}")
else:
    print(prof.key_averages(group_by_input_shape="--shapes" in sys.argv).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=50))
