"""Dev experiment: does channels_last help the (non-hot-path) convolution side of the step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffusion_finetuning_amd.trainer import LoraTrainer
dev = torch.device("cuda", 0)
for cl in (False, True):
    unet = bench.build_model(dev, torch.float16, 4)
    if cl:
        unet = unet.to(memory_format=torch.channels_last)
    tr = LoraTrainer(unet, lr=1e-4)
    data = bench.synthetic_steps(8, 4, 64, 0, 1, dev)
    for i in range(3): tr.step(*data[i])
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(3, 8): tr.step(*data[i])
    torch.cuda.synchronize(); print("channels_last", cl, (time.perf_counter() - t) / 5 * 1e3, "ms/step", flush=True)
    del tr, unet; torch.cuda.empty_cache()
