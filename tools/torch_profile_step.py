"""Dev tool: torch.profiler view of one train step (which aten ops own the GPU time outside the hot path)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(*data[3]); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
