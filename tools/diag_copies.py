"""Dev tool: how often does the autograd glue copy (non-contiguous / dtype-mismatched) operands per step?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffusion_finetuning_amd import ops
from diffusion_finetuning_amd.trainer import LoraTrainer
cnt = collections.Counter()
orig_fwd, orig_bwd = ops._LoraLinearFn.forward, ops._LoraLinearFn.backward
def fwd(ctx, x, *a):
    x2 = x.reshape(-1, x.shape[-1])
    cnt["fwd_calls"] += 1
    cnt["fwd_x_noncontig"] += int(not x2.is_contiguous())
    cnt["fwd_x_dtype_cast"] += int(x.dtype != a[2].dtype)
    return orig_fwd(ctx, x, *a)
def bwd(ctx, dy):
    cnt["bwd_calls"] += 1
    d2 = dy.reshape(-1, dy.shape[-1])
    cnt["bwd_dy_noncontig"] += int(not d2.is_contiguous())
    if not d2.is_contiguous(): cnt[f"  shape{tuple(dy.shape)} stride{tuple(dy.stride())}"] += 1
    return orig_bwd(ctx, dy)
ops._LoraLinearFn.forward = staticmethod(fwd); ops._LoraLinearFn.backward = staticmethod(bwd)
dev = torch.device("cuda", 0)
unet = bench.build_model(dev, torch.float16, 4)
tr = LoraTrainer(unet, lr=1e-4)
data = bench.synthetic_steps(2, 4, 64, 0, 1, dev)
tr.step(*data[0]); cnt.clear(); tr.step(*data[1]); torch.cuda.synchronize()
for k, v in sorted(cnt.items()): print(k, v)
