"""Dev experiment: capture add_noise → UNet forward → loss → backward of one train step in a hipGraph and replay it;
partial-sum fold, exchange and optimizer stay eager (no collective inside the graph).  Prints eager vs graph ms/step
and checks that both leave the same LoRA state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from diffusion_finetuning_amd import _native as nat
from diffusion_finetuning_amd.trainer import LoraTrainer, flat_lora_state
dev = torch.device("cuda", 0)
K = 10
data = bench.synthetic_steps(K + 4, 4, 64, 0, 1, dev)

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n

# ---- eager
unet = bench.build_model(dev, torch.float16, 4)
tr = LoraTrainer(unet, lr=1e-4)
for i in range(4): tr.step(*data[i])
t_eager = timed(lambda i: tr.step(*data[4 + i]), K)
state_eager = flat_lora_state(unet).clone()

# ---- graph
unet = bench.build_model(dev, torch.float16, 4)
tr = LoraTrainer(unet, lr=1e-4)
lat, noise, ts, ehs = (t.clone() for t in data[0])
ehs16 = ehs.to(tr.dtype)
out = {}
def body():
    noisy, target = nat.ddpm_add_noise(lat, noise, ts, tr.sqrt_acp, tr.sqrt_1macp, tr.dtype, tr.v_prediction)
    pred = tr.unet(noisy, ts, ehs16).sample
    pred_c = pred if pred.is_contiguous() else pred.contiguous()
    loss, dpred = nat.ddpm_mse_fwd_bwd(pred_c, target, None, pred.shape[0], 0, 1.0, tr.loss_scale)
    pred_c.backward(dpred)
    out["loss"] = loss
def pre(i):
    for dst, src in zip((lat, noise, ts), data[i][:3]): dst.copy_(src)
    ehs16.copy_(data[i][3])
    tr.slab.zero_grad(); tr.slab.repack()
def post():
    for s in tr.slab._sinks: s.ran = 0
    tr.exchange.finish(); tr.opt.step(grad_mul=1.0 / (tr.world * tr.loss_scale)); tr.slab.repack()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(4):
        pre(i); body(); post()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
# the eager run above took the same 4 warm-up steps: states must agree here
g = torch.cuda.CUDAGraph()
pre(4)
with torch.cuda.graph(g):
    body()
print("captured", flush=True)
def gstep(i):
    pre(4 + i); g.replay(); post()
# the capture itself did not execute step 4; replay it and the following ones
t_graph = timed(gstep, K)
state_graph = flat_lora_state(unet)
err = ((state_graph - state_eager).norm() / state_eager.norm()).item()
print(f"eager {t_eager:.2f} ms/step   graph {t_graph:.2f} ms/step   LoRA state rel diff {err:.2e}   loss {out['loss'].item():.4f}")
