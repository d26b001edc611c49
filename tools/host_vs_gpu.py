"""Dev tool: is the train step host-bound?  Times the ENQUEUE of K steps (no sync inside) against the wall time
until the GPU has drained them, and the GPU busy time from events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from diffusion_finetuning_amd.trainer import LoraTrainer
dev = torch.device("cuda", 0)
unet = bench.build_model(dev, torch.float16, 4)
tr = LoraTrainer(unet, lr=1e-4)
K = 10
data = bench.synthetic_steps(K + 4, 4, 64, 0, 1, dev)
for i in range(4): tr.step(*data[i])
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for i in range(K): tr.step(*data[4 + i])
t1 = time.perf_counter(); e1.record()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.2f} ms/step   drained {1e3*(t2-t0)/K:.2f} ms/step   gpu(events) {e0.elapsed_time(e1)/K:.2f} ms/step")
# forward-only / backward split of the host time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(3): tr.step(*data[4 + i])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
