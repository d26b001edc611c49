"""Dev tool: torch.profiler view of one train step (which aten ops own the GPU time outside the hot path).
--shapes: group by input shapes; --stack OP: print the Python call sites of one aten op (e.g. aten::copy_)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=stack_op is not None) as prof:
    tr.step(*data[3]); torch.cuda.synchronize()
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
