import sys, os, itertools
sys.path.insert(0, "/root/repo")
import torch
from tests.test_gpu_groups import _train, _tiny64, _warm
from tests.conftest import rel_err
from oracle import lora_oracle as orc
for r in (4, 8, 16):
    ref = _tiny64(); ref_params, _ = orc.inject(ref, r=r); _warm(ref_params)
    init = orc.flat_params(ref_params).clone()
    ref_losses = orc.train_steps(ref, ref_params, 4, 2, 8, 6, 64, lr=1e-3)
    want = orc.flat_params(ref_params)
    _, g, lg = _train(True, True, dtype=torch.float16, r=r)
    _, u, lu = _train(False, True, dtype=torch.float16, r=r)
    _, g32, _ = _train(True, True, dtype=torch.float32, r=r)
    print(f"r={r}: grouped-vs-ungrouped {rel_err(g,u):.2e} | grouped-vs-oracle {rel_err(g,want):.2e} | ungrouped-vs-oracle {rel_err(u,want):.2e} | f32 grouped-vs-oracle {rel_err(g32,want):.2e} | updates: g {rel_err(g-init,want-init):.2e} u {rel_err(u-init,want-init):.2e} g-vs-u {rel_err(g-init,u-init):.2e}", flush=True)
