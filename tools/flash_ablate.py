"""Dev tool: device time of the flash backward kernels alone (dispatch events), for ablation builds of the library
(tools/build_variant.sh abl<N> -DFLASH_ABL=<N>; DFA_LIB_PATH selects one).  Results of an ablation build are wrong: timing only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_finetuning_amd import _native as nat
dev = "cuda"
torch.manual_seed(0)
for (B, T, H, d) in [(4, 4096, 8, 40), (4, 1024, 8, 80)]:
    q = torch.randn(B, T, H * d, device=dev).half(); k = torch.randn_like(q); v = torch.randn_like(q); go = torch.randn_like(q)
    o, lse = nat.attn_flash_fwd(q, k, v, H, d ** -0.5)
    for _ in range(3): nat.attn_flash_bwd(q, k, v, o, go, lse, H, d ** -0.5)
    torch.cuda.synchronize()
    nat.prof_enable(200)
    for _ in range(20): nat.attn_flash_bwd(q, k, v, o, go, lse, H, d ** -0.5)
    torch.cuda.synchronize()
    res = nat.prof_collect(); nat.prof_enable(0)
    print(f"T={T} d={d} lib={os.path.basename(nat.library_path())}: " + "  ".join(f"{k_[:24]} {1e3 * v_['ms'] / v_['launches']:.1f} us" for k_, v_ in res.items()), flush=True)
