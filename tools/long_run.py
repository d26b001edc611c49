"""Dev check: a few hundred train steps of the SD1.5-shaped model on ONE fixed batch (so the loss must fall), with the
hipGraph step, both 16-bit dtypes: no NaN/Inf, no loss-scale overflow, loss decreasing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from diffusion_finetuning_amd.trainer import LoraTrainer
dev = torch.device("cuda", 0)
for dtype in (torch.float16, torch.bfloat16):
    unet = bench.build_model(dev, dtype, 4)
    tr = LoraTrainer(unet, lr=2e-4, capture_graph=True)
    data = bench.synthetic_steps(1, 4, 64, 0, 1, dev)[0]
    losses = []
    for i in range(300):
        l = tr.step(*data)
        if i % 50 == 0 or i == 299: losses.append(round(l.item(), 5))
    print(str(dtype)[6:], "loss", losses, "overflow", tr.opt.overflowed(), "grad norm", round(tr.opt.grad_norm(), 4),
          "finite", bool(torch.isfinite(tr.slab.params).all()), flush=True)
