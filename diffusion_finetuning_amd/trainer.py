"""Step harness for the LoRA hot path: flat LoRA slab, fused loss, data-parallel exchange, fused clip+AdamW.

Reproduces the per-step semantics of training_scripts/train_lora_dreambooth.py:811-888 (and the PTI tuning
loop lora_diffusion/cli_lora_pti.py:438-451) on synthetic latents:

    noisy = add_noise(latents, noise, t) ; pred = unet(noisy, t, ctx).sample ; target = noise | velocity
    loss  = mse (+ prior preservation | mask) ; backward ; [DDP mean all-reduce of LoRA grads]
    clip_grad_norm_(·, max_grad_norm) ; AdamW.step ; zero_grad

MI355X-first layout: every LoRA factor of the model lives in ONE fp32 slab in enumeration order
[up0, down0, up1, down1, ...] (the order of `inject_trainable_lora`'s return value and of the `.pt` file), with
matching slabs for gradients and both Adam moments.  The backward kernels accumulate straight into the gradient
slab, which is also the RCCL send/receive buffer — no packing, one or two collectives per step instead of 288 —
and the optimizer is two launches over the slab.  Data parallelism is one process per GPU over
`torch.distributed` (backend "nccl" = RCCL over xGMI); only the LoRA gradients (≈5 MB at rank 4) ever cross GPUs.
"""
import math
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _native as nat
from .core import LoraInjectedLinear


def ddpm_tables(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, device="cpu"):
    """sqrt(ᾱ_t), sqrt(1-ᾱ_t) of the SD "scaled_linear" DDPM schedule (the constants come from the model's hub
    config, which is not part of the reference repo)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    return acp.sqrt().to(device), (1.0 - acp).sqrt().to(device)


def lora_layers(model: nn.Module) -> List[LoraInjectedLinear]:
    """All LoraInjectedLinear modules in `model.modules()` order — for an injected model this is the
    enumeration order of lora.py:78-114, because every target is visited under its first matching ancestor."""
    return [m for m in model.modules() if isinstance(m, LoraInjectedLinear)]


class LoraSlab:
    """Re-homes every LoRA factor of `models` into one flat fp32 parameter slab (+ gradient slab).

    The nn.Parameter objects stay the same (optimizers, generators and state_dict keep working); only their
    storage moves.  Each layer gets `_dfa_grad_sink = (grad_down_view, grad_up_view)` so the backward kernel
    accumulates in place, and `.grad` of each Parameter is a view of the gradient slab."""

    GRAD_BLOCKS = 128

    def __init__(self, models: Sequence[nn.Module]):
        self._sinks = []
        self.layers: List[LoraInjectedLinear] = []
        self.model_ranges: List[Tuple[int, int]] = []
        for model in models:
            start = sum(l.lora_up.weight.numel() + l.lora_down.weight.numel() for l in self.layers)
            self.layers += lora_layers(model)
            end = sum(l.lora_up.weight.numel() + l.lora_down.weight.numel() for l in self.layers)
            self.model_ranges.append((start, end))
        if not self.layers:
            raise ValueError("No lora injected.")
        device = self.layers[0].lora_up.weight.device
        if device.type != "cuda":
            raise RuntimeError("LoraSlab: the model must be on the HIP device")
        total = sum(l.lora_up.weight.numel() + l.lora_down.weight.numel() for l in self.layers)
        pad = (-total) % 4  # keep 16-byte granularity for vector loads
        self.numel = total
        self.stride = total + pad
        self.params = torch.zeros(total + pad, dtype=torch.float32, device=device)
        self.grads = torch.zeros(total + pad, dtype=torch.float32, device=device)
        # row-block partial sums of the factor gradients, [GRAD_BLOCKS][slab]: written by the backward kernels
        # with plain stores, summed in block order by ONE launch per step (deterministic, no atomics)
        self.partials = torch.zeros((self.GRAD_BLOCKS, total + pad), dtype=torch.float32, device=device)
        self.offsets = []
        off = 0
        for layer in self.layers:
            views = {}
            for attr in ("lora_up", "lora_down"):  # file order: up then down
                p = getattr(layer, attr).weight
                n = p.numel()
                pv = self.params[off:off + n].view(p.shape)
                pv.copy_(p.detach().float())
                p.data = pv
                gv = self.grads[off:off + n].view(p.shape)
                p.grad = gv
                views[attr] = gv
                self.offsets.append((off, n))
                off += n
            up_off, down_off = self.offsets[-2][0], self.offsets[-1][0]
            sink = GradSink(self.partials, up_off, down_off, down_off + self.offsets[-1][1], self.stride, self.GRAD_BLOCKS)
            self._sinks.append(sink)
            layer.__dict__["_dfa_grad_sink"] = sink

    def enable_packed(self, dtype: torch.dtype):
        """Allocates the packed-factor slab (Apack 32·K + Bpack 32·N per layer in the compute dtype, both
        orientations: lora_hip.h) and the device table for the one-launch re-pack; each layer gets
        `_dfa_packed = (Apack, Bpack)` views."""
        rows = []
        off = 0
        self._packed_layers = []
        for i, layer in enumerate(self.layers):
            r, K = layer.lora_down.weight.shape
            N = layer.lora_up.weight.shape[0]
            if r > 16:
                continue
            up_off, down_off = self.offsets[2 * i][0], self.offsets[2 * i + 1][0]
            rows.append([down_off, up_off, K, N, r, off, off + 32 * K, 0])
            self._packed_layers.append((layer, off, K, N))
            off += 32 * (K + N)
        if not rows:
            self.packed = None
            return
        self.packed = torch.zeros(off, dtype=dtype, device=self.params.device)
        self._pack_table = torch.tensor(rows, dtype=torch.int64, device=self.params.device)
        self._pack_maxlen = max(max(r_[2], r_[3]) for r_ in rows)
        for layer, o, K, N in self._packed_layers:
            layer.__dict__["_dfa_packed"] = (self.packed[o:o + 32 * K], self.packed[o + 32 * K:o + 32 * (K + N)])
        self.repack()

    def repack(self):
        """Refresh every packed factor from the fp32 master slab (one launch)."""
        if getattr(self, "packed", None) is not None:
            nat.lora_pack_factors_batched(self._pack_table, len(self._packed_layers), self._pack_maxlen, self.params,
                                          self.packed)

    def zero_grad(self):
        self.grads.zero_()
        for sink in self._sinks:
            sink.ran = 0

    def check_all_layers_ran(self):
        """A layer that did not run backward in this pass still holds an older pass's partials: clear those
        slices before they are reduced (never the case for a UNet step)."""
        for sink in self._sinks:
            if sink.ran == 0:
                sink.clear(0)
            elif sink.ran > 1:
                raise RuntimeError("a LoRA layer ran backward more than once in one pass (shared module?): "
                                   "the partial-sum layout holds one pass per layer")
            sink.ran = 0

    def reduce_range(self, a: int, b: int):
        """grads[a:b] += Σ_blocks partials[:, a:b] (block order, deterministic)."""
        if a % 4:
            raise RuntimeError("slab range must start on a 16-byte boundary")
        nat.lora_reduce_partials(self.partials[0, a:], self.stride, self.GRAD_BLOCKS, self.grads[a:], b - a, True)

    def detach_sinks(self):
        for layer in self.layers:
            layer.__dict__.pop("_dfa_grad_sink", None)
            layer.__dict__.pop("_dfa_packed", None)

    def range_of(self, module: nn.Module) -> Tuple[int, int]:
        """[start, end) of the slab covering the LoRA layers under `module` (must be contiguous)."""
        inside = {id(l) for l in lora_layers(module)}
        idx = [i for i, l in enumerate(self.layers) if id(l) in inside]
        if not idx:
            return (0, 0)
        assert idx == list(range(idx[0], idx[-1] + 1)), "layers under the module are not contiguous in the slab"
        return (self.offsets[2 * idx[0]][0], self.offsets[2 * idx[-1] + 1][0] + self.offsets[2 * idx[-1] + 1][1])


class GradSink:
    """Where one layer's factor-gradient kernel stores its row-block partial sums inside the model-wide
    [GRAD_BLOCKS][slab] buffer.  The number of row blocks follows M (≈128 rows per block, so small layers do not
    write — and the fold does not read — mostly-empty blocks); blocks a layer stops using are zeroed once."""

    def __init__(self, partials, up_off, down_off, end, stride, max_blocks):
        self.partials, self.begin, self.end = partials, up_off, end
        self.ga_part = partials[0, down_off:]
        self.gb_part = partials[0, up_off:]
        self.stride, self.max_blocks = stride, max_blocks
        self.used = 0   # blocks holding data from the last pass
        self.ran = 0

    def blocks_for(self, M: int) -> int:
        nb = max(min(4, self.max_blocks), min(self.max_blocks, M // 32))
        if nb < self.used:
            self.partials[nb:self.used, self.begin:self.end].zero_()
        self.used = nb
        self.ran += 1
        return nb

    def clear(self, keep: int):
        if self.used > keep:
            self.partials[keep:self.used, self.begin:self.end].zero_()
            self.used = keep


class FusedClipAdamW:
    """clip_grad_norm_ + torch.optim.AdamW over slab ranges, each range with its own lr / weight decay
    (the reference builds one param group for the UNet and one for the text encoder,
    train_lora_dreambooth.py:659-676)."""

    def __init__(self, slab: LoraSlab, groups: Sequence[dict], betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0):
        self.slab = slab
        self.groups = [dict(g) for g in groups]  # {"range": (a,b), "lr": .., "weight_decay": ..}
        self.betas, self.eps, self.max_grad_norm = betas, eps, max_grad_norm
        self.exp_avg = torch.zeros_like(slab.params)
        self.exp_avg_sq = torch.zeros_like(slab.params)
        self.norm = torch.zeros(4, dtype=torch.float32, device=slab.params.device)
        self.step_count = 0

    def step(self, grad_mul: float = 1.0):
        """One optimizer step on the current gradient slab; `grad_mul` = 1/(world_size·loss_scale)."""
        s = self.slab
        self.step_count += 1
        nat.lora_grad_sqnorm(s.grads[: s.numel], grad_mul, self.norm)
        for g in self.groups:
            a, b = g["range"]
            if b <= a:
                continue
            nat.lora_adamw_step(s.params[a:b], s.grads[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], self.norm,
                                grad_mul, self.max_grad_norm, g["lr"], self.betas[0], self.betas[1], self.eps,
                                g.get("weight_decay", 1e-2), self.step_count)

    def grad_norm(self) -> float:
        """Total (pre-clip) gradient L2 norm of the last step (host sync)."""
        return math.sqrt(float(self.norm[0].item()))

    def overflowed(self) -> bool:
        return bool(self.norm[1].item() != 0.0)


class SlabExchange:
    """Synchronous data-parallel exchange of the flat gradient slab: SUM all-reduce in at most two buckets
    (the mean's 1/world is folded into the optimizer's grad_mul).  The reference gets the same effect
    implicitly from DDP inside accelerator.backward (train_lora_dreambooth.py:744-757,877).  Device-agnostic:
    RCCL ("nccl") on the GPUs, gloo in the CPU tests.

    `prepare(a, b)` (optional) is called for every slab range right before it is sent — the trainer uses it to
    fold the row-block partial sums of that range into the gradient slab — and also when world == 1."""

    def __init__(self, grads: torch.Tensor, numel: int, process_group=None, prepare=None, always_reduce=False):
        self.grads, self.numel, self.pg, self.prepare = grads, numel, process_group, prepare
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # always_reduce: run the bucketed all-reduce path even on a 1-rank group (used to test the RCCL path
        # on a single GPU)
        self.active = self.world > 1 or (always_reduce and dist.is_available() and dist.is_initialized())
        self.early_range: Optional[Tuple[int, int]] = None
        self._armed = False
        self._early_sent = False
        self._pending = []

    def arm(self):
        """Call before each backward pass; launch_early() then fires at most once."""
        self._armed = self.active and self.early_range is not None
        self._early_sent = False

    def _send(self, a: int, b: int):
        if b <= a:
            return
        if self.prepare is not None:
            self.prepare(a, b)
        if self.active:
            self._pending.append(dist.all_reduce(self.grads[a:b], group=self.pg, async_op=True))

    def launch_early(self):
        """Start reducing the early bucket (its gradients are final) while backward continues."""
        if self._armed:
            self._armed = False
            self._early_sent = True
            self._send(*self.early_range)

    def finish(self):
        """Send whatever has not been sent yet and wait for every bucket."""
        n = self.numel
        if self._early_sent:
            a, b = self.early_range
            self._send(0, a)
            self._send(b, n)
        else:
            self._send(0, n)
        for w in self._pending:
            w.wait()
        self._pending = []
        self._armed = False
        self._early_sent = False


class LoraTrainer:
    """One object per process (= per GPU).  `step()` runs one full training step and returns the loss tensor
    (no host sync unless the caller reads it)."""

    def __init__(self, unet: nn.Module, text_encoder: Optional[nn.Module] = None, lr=1e-4, lr_text=5e-6,
                 weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, loss_scale: Optional[float] = None,
                 v_prediction=False, process_group=None, always_reduce=False, capture_graph=False):
        """capture_graph: record add_noise → UNet forward → loss → backward of a step once into a hipGraph and
        replay it on later steps with the same shapes (inputs are copied into static buffers).  The partial-sum fold,
        the gradient exchange and the optimizer stay outside the graph, so no collective is ever captured.  Only the
        plain UNet step is eligible (no text-encoder LoRA, no mask); anything else runs eagerly."""
        self.unet, self.text_encoder = unet, text_encoder
        self.capture_graph = bool(capture_graph)
        self._graph = None
        models = [unet] + ([text_encoder] if text_encoder is not None and lora_layers(text_encoder) else [])
        self.slab = LoraSlab(models)
        groups = [{"range": self.slab.model_ranges[0], "lr": lr, "weight_decay": weight_decay}]
        if len(models) > 1:
            groups.append({"range": self.slab.model_ranges[1], "lr": lr_text, "weight_decay": weight_decay})
        self.opt = FusedClipAdamW(self.slab, groups, betas, eps, max_grad_norm)
        self.device = self.slab.params.device
        self.dtype = next(p for p in unet.parameters() if p.dim() == 4).dtype  # conv weight dtype = compute dtype
        self.loss_scale = float(loss_scale) if loss_scale is not None else (1024.0 if self.dtype == torch.float16 else 1.0)
        self.slab.enable_packed(self.dtype)
        self.v_prediction = v_prediction
        self.sqrt_acp, self.sqrt_1macp = ddpm_tables(device=self.device)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.exchange = SlabExchange(self.slab.grads, self.slab.numel, process_group, prepare=self.slab.reduce_range,
                                     always_reduce=always_reduce)
        if self.exchange.active:
            self._broadcast_initial_state()
            self._install_bucket_hook()

    # -- data parallel -------------------------------------------------------------------------
    def _broadcast_initial_state(self):
        """DDP construction broadcasts module state from rank 0 (train_lora_dreambooth.py:751-757); the frozen
        base is loaded identically everywhere, so only the LoRA slab travels."""
        dist.broadcast(self.slab.params, src=0, group=self.pg)

    def _install_bucket_hook(self):
        """Two buckets in backward-completion order.  Enumeration order is down, up, mid; backward finishes the
        up blocks first, then mid, then the down blocks — so [up|mid] is a contiguous early bucket whose
        all-reduce overlaps the rest of the backward pass."""
        mid = getattr(self.unet, "mid_block", None)
        ups = getattr(self.unet, "up_blocks", None)
        if mid is None or ups is None:
            return
        a0, a1 = self.slab.range_of(ups)
        b0, b1 = self.slab.range_of(mid)
        if a1 != b0 or a1 <= a0:
            return
        self.exchange.early_range = (a0, b1)
        mid.register_full_backward_hook(lambda module, gin, gout: self.exchange.launch_early())

    # -- one step ---------------------------------------------------------------------------------
    def step(self, latents, noise, timesteps, encoder_hidden_states, *, with_prior_preservation=False,
             prior_loss_weight=1.0, mask=None, seed: Optional[int] = None):
        """latents fp32 [B,4,h,w] on the device, encoder_hidden_states [B,L,D].  Either pass `noise` (fp32, like
        latents) and `timesteps` (int64 [B]) — the caller drew them, as the reference does — or pass None for both and
        a `seed`: the step then draws them on the device (Philox keyed by (seed, optimizer step), identical on
        every rank) inside the prologue kernel."""
        if self.capture_graph and mask is None and self.text_encoder is None:
            return self._step_graph(latents, noise, timesteps, encoder_hidden_states, with_prior_preservation,
                                    prior_loss_weight, seed)
        return self._step_eager(latents, noise, timesteps, encoder_hidden_states, with_prior_preservation,
                                prior_loss_weight, mask, seed, early_bucket=True)

    def _step_eager(self, latents, noise, timesteps, encoder_hidden_states, with_prior_preservation, prior_loss_weight,
                    mask, seed, early_bucket):
        """Host-launched step.  early_bucket=False sends the slab in ONE all-reduce after backward — the collective
        sequence of a hipGraph step — so a rank whose graph recording failed stays matched with ranks that replay."""
        self.slab.zero_grad()
        self.slab.repack()  # packed compute-dtype factors follow the fp32 masters (also after external edits)
        if noise is None:
            if seed is None:
                raise ValueError("pass noise and timesteps, or a seed for the on-device draw")
            noisy, target, timesteps = nat.ddpm_noise_prologue(latents, self.sqrt_acp, self.sqrt_1macp, self.dtype, seed,
                                                               self.opt.step_count, self.v_prediction)
        else:
            noisy, target = nat.ddpm_add_noise(latents, noise, timesteps, self.sqrt_acp, self.sqrt_1macp, self.dtype,
                                               self.v_prediction)
        if early_bucket:
            self.exchange.arm()
        pred = self.unet(noisy, timesteps, encoder_hidden_states.to(self.dtype)).sample
        rows = pred.shape[0]
        n_inst, n_prior = (rows // 2, rows // 2) if with_prior_preservation else (rows, 0)
        m = None
        if mask is not None:
            raw = mask.to(self.device).reshape(rows, 1, pred.shape[2] * 8, pred.shape[3] * 8).float().contiguous()
            m = nat.lora_mask_prepare(raw, pred.shape[2], pred.shape[3])
        pred_c = pred if pred.is_contiguous() else pred.contiguous()
        loss, dpred = nat.ddpm_mse_fwd_bwd(pred_c, target, m, n_inst, n_prior, prior_loss_weight, self.loss_scale)
        pred_c.backward(dpred)
        self.slab.check_all_layers_ran()
        self.exchange.finish()
        self.opt.step(grad_mul=1.0 / (self.world * self.loss_scale))
        self.slab.repack()  # forwards outside step() (sampling, evaluation, saving merged weights) see the new factors
        return loss


    # -- the same step with forward+backward replayed from a hipGraph -----------------------------------
    def _graph_body(self, st):
        """What is captured: reads only static buffers, leaves the factor-gradient partials and the loss behind."""
        if st["draw"]:
            noisy, target = st["noisy"], st["target"]
        else:
            noisy, target = nat.ddpm_add_noise(st["latents"], st["noise"], st["timesteps"], self.sqrt_acp,
                                               self.sqrt_1macp, self.dtype, self.v_prediction)
        pred = self.unet(noisy, st["timesteps"], st["ehs"]).sample
        rows = pred.shape[0]
        n_inst, n_prior = (rows // 2, rows // 2) if st["prior"] else (rows, 0)
        pred_c = pred if pred.is_contiguous() else pred.contiguous()
        loss, dpred = nat.ddpm_mse_fwd_bwd(pred_c, target, None, n_inst, n_prior, st["prior_weight"], self.loss_scale)
        pred_c.backward(dpred)
        st["loss"] = loss

    def _graph_inputs(self, st, latents, noise, timesteps, ehs, seed):
        if st["draw"]:
            if seed is None:
                raise ValueError("pass noise and timesteps, or a seed for the on-device draw")
            noisy, target, t = nat.ddpm_noise_prologue(latents, self.sqrt_acp, self.sqrt_1macp, self.dtype, seed,
                                                       self.opt.step_count, self.v_prediction)
            st["noisy"].copy_(noisy)
            st["target"].copy_(target)
            st["timesteps"].copy_(t)
        else:
            st["latents"].copy_(latents)
            st["noise"].copy_(noise)
            st["timesteps"].copy_(timesteps)
        st["ehs"].copy_(ehs)  # casts to the compute dtype
        self.slab.zero_grad()
        self.slab.repack()

    def _step_graph(self, latents, noise, timesteps, ehs, prior, prior_weight, seed):
        key = (tuple(latents.shape), tuple(ehs.shape), bool(prior), float(prior_weight), noise is None)
        st = self._graph
        if st is None or st["key"] != key:
            st = {"key": key, "draw": noise is None, "prior": bool(prior), "prior_weight": float(prior_weight),
                  "latents": torch.empty_like(latents, dtype=torch.float32),
                  "noise": torch.empty_like(latents, dtype=torch.float32),
                  "timesteps": torch.empty(latents.shape[0], dtype=torch.int64, device=self.device),
                  "noisy": torch.empty_like(latents, dtype=self.dtype), "target": torch.empty_like(latents, dtype=self.dtype),
                  "ehs": torch.empty_like(ehs, dtype=self.dtype), "graph": None}
            self._graph_inputs(st, latents, noise, timesteps, ehs, seed)
            try:
                # warm up on a side stream (solver searches, lazy initialisation, allocator), then record
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._graph_body(st)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._graph_body(st)
                st["graph"] = g
            except Exception as exc:  # keep training: this trainer falls back to eager steps for good
                import warnings

                warnings.warn(f"LoraTrainer: hipGraph capture failed ({exc!r}); continuing with eager steps")
                self.capture_graph, self._graph = False, None
                # THIS step still exchanges like a graph step (one all-reduce): peers may have recorded fine
                return self._step_eager(latents, noise, timesteps, ehs, prior, prior_weight, None, seed,
                                        early_bucket=False)
            self._graph = st
            # the warm-up passes left valid partial sums of THIS step's inputs, but replay once so that every step
            # (including the first) is produced by the same recorded kernels
        else:
            self._graph_inputs(st, latents, noise, timesteps, ehs, seed)
        st["graph"].replay()
        for sink in self.slab._sinks:  # the Python backward fronts do not run on replay: nothing to check
            sink.ran = 0
        self.exchange.finish()
        self.opt.step(grad_mul=1.0 / (self.world * self.loss_scale))
        self.slab.repack()
        return st["loss"].clone()


def flat_lora_state(model: nn.Module, targets=None) -> torch.Tensor:
    """[up0, down0, up1, ...] flattened — the `.pt` file order."""
    return torch.cat([t.detach().float().reshape(-1) for l in lora_layers(model)
                      for t in (l.lora_up.weight, l.lora_down.weight)])
