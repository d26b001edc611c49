"""Dev tool: where does the unchanged-trainer route (bench.py --drop-in) spend its step?  Wall time per step against the sum of
GPU kernel time (torch.profiler), and the top ops by self GPU time and by self CPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import itertools
import torch, torch.nn.functional as F
import bench
import diffusion_finetuning_amd as dfa
from diffusion_finetuning_amd.attention import set_use_memory_efficient_attention_xformers
from diffusion_finetuning_amd.trainer import ddpm_tables
from harness.unet import UNet2DConditionModel, sd15_config
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
torch.manual_seed(0)
with torch.device(dev):
    unet = UNet2DConditionModel(sd15_config())
unet.requires_grad_(False)
params, _ = dfa.inject_trainable_lora(unet, r=4)
set_use_memory_efficient_attention_xformers(unet, True)
plist = list(itertools.chain(*params))
opt = torch.optim.AdamW(plist, lr=1e-4)
scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
sa, sb = ddpm_tables(device=dev)
data = bench.synthetic_steps(6, 4, 64, 0, 1, dev)

def step(i):
    lat, _, _, ctx = data[i]
    noise = torch.randn_like(lat); t = torch.randint(0, 1000, (lat.shape[0],), device=dev)
    noisy = sa[t].view(-1, 1, 1, 1) * lat + sb[t].view(-1, 1, 1, 1) * noise
    with torch.autocast("cuda", dtype=torch.float16):
        pred = unet(noisy, t, ctx).sample
    loss = F.mse_loss(pred.float(), noise.float())
    scaler.scale(loss).backward(); scaler.unscale_(opt)
    torch.nn.utils.clip_grad_norm_(plist, 1.0); scaler.step(opt); scaler.update(); opt.zero_grad()

for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3, 6): step(i)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(5); torch.cuda.synchronize()
ka = prof.key_averages()
gpu = sum(e.self_device_time_total for e in ka) / 1e3
print(f"wall {1e3 * wall:.1f} ms/step; GPU kernel time {gpu:.1f} ms in the profiled step; {sum(e.count for e in ka if e.self_device_time_total > 0)} device ops")
print(ka.table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=60))
print(ka.table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
