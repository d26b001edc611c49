"""Dev experiment: ms/step of the hipGraph step with a host sync after every step (what a host-staged collective such
as gloo does), alone and with a second process on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from diffusion_finetuning_amd.trainer import LoraTrainer
dev = torch.device("cuda", 0)
graph = "--no-graph" not in sys.argv
unet = bench.build_model(dev, torch.float16, 4)
tr = LoraTrainer(unet, lr=1e-4, capture_graph=graph)
data = bench.synthetic_steps(9, 4, 64, 0, 1, dev)
for i in range(3): tr.step(*data[i])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3, 9):
    tr.step(*data[i]); torch.cuda.synchronize()
print(f"graph={graph} per-step sync: {1e3 * (time.perf_counter() - t0) / 6:.1f} ms/step", flush=True)
