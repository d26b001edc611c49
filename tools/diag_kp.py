"""Dev diagnostic: lora_gemm_parts backward (KP = 3) at tiny contractions, run to run and against float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_finetuning_amd import _native as nat
DEV = "cuda"
def case(M, K, N, r, dtype=torch.float16, G=3):
    g = torch.Generator().manual_seed(7)
    Ws = [((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dtype) for _ in range(G)]
    As = [(torch.randn(r, K, generator=g) / r) for _ in range(G)]
    Bs = [(torch.randn(N, r, generator=g) * 0.05) for _ in range(G)]
    dY = torch.randn(M, G * N, generator=g).to(dtype).to(DEV)
    params = torch.cat([t.reshape(-1) for pair in zip(Bs, As) for t in pair]).to(DEV)
    fa, qb, fb, qa = 0, 16 * G * K, 16 * G * K + 16 * G * N, 16 * G * K + 32 * G * N
    rows, po = [], 0
    for i in range(G):
        up_off, down_off = po, po + N * r
        po += N * r + r * K
        rows.append([down_off, 0, K, r, fa + i * 16 * K, K, qa + i * 16 * K, 16])
        rows.append([up_off, 1, N, r, fb + i * N, G * N, qb + i * N * 16, 16])
    packed = torch.zeros(32 * G * (K + N), dtype=dtype, device=DEV)
    nat.lora_pack_items(torch.tensor(rows, dtype=torch.int64).to(DEV), len(rows), max(K, N), params, packed)
    Fb, Qa = packed[fb:qa], packed[qa:]
    Wt = torch.cat(Ws).to(DEV).t().contiguous()
    ref = torch.zeros(M, K, dtype=torch.float64)
    for i in range(G):
        dy_i = dY[:, i * N:(i + 1) * N].double().cpu()
        a, b = As[i].to(dtype).double(), Bs[i].to(dtype).double()
        ref += dy_i @ Ws[i].double() + 0.7 * (dy_i @ b) @ a
    outs = []
    for _ in range(4):
        dX = torch.full((M, K), float("nan"), dtype=dtype, device=DEV)
        U = torch.full((M, G * r), float("nan"), device=DEV)
        assert nat.lora_gemm_parts(dY, Wt, None, Fb, Qa, dX, U, G * r, M, G * N, K, r, G, True, 0.7)
        torch.cuda.synchronize()
        outs.append(dX.double().cpu())
    errs = [((o - ref).norm() / ref.norm()).item() for o in outs]
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    d = (outs[0] - ref).abs()
    bad_rows = (d.max(dim=1).values > 0.05 * ref.abs().max()).nonzero().flatten().tolist()
    bad_cols = (d.max(dim=0).values > 0.05 * ref.abs().max()).nonzero().flatten().tolist()
    print(f"M={M} K={K} N={N} r={r}: errs {[f'{e:.2e}' for e in errs]} identical runs {same}; bad rows {bad_rows[:8]}..({len(bad_rows)}) bad cols {bad_cols[:8]}..({len(bad_cols)})", flush=True)
for shp in [(100, 64, 64, 6), (64, 64, 64, 6), (128, 64, 64, 8), (100, 128, 64, 6), (100, 64, 128, 6), (100, 128, 128, 8), (100, 192, 64, 6), (1024, 64, 64, 8)]:
    case(*shp)
