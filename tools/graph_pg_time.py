"""Dev experiment: ms/step of the hipGraph step followed by the slab all-reduce on a 1-rank process group
(backend from argv: nccl = RCCL, or gloo), against the eager step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import torch.distributed as dist
from diffusion_finetuning_amd.trainer import LoraTrainer
backend = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dev = torch.device("cuda", 0)
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group(backend, rank=0, world_size=1)
data = bench.synthetic_steps(9, 4, 64, 0, 1, dev)
for graph in (False, True):
    unet = bench.build_model(dev, torch.float16, 4)
    tr = LoraTrainer(unet, lr=1e-4, capture_graph=graph, always_reduce=True)
    for i in range(3): tr.step(*data[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3, 9): tr.step(*data[i])
    torch.cuda.synchronize()
    print(f"{backend} graph={graph}: {1e3 * (time.perf_counter() - t0) / 6:.1f} ms/step", flush=True)
dist.destroy_process_group()
