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
import os
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


SCHEDULER_NAMES = ("linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup")


def survey_bytes_flops(layers, esize: int = 2, contract: bool = False):
    """SURVEY §8(d), "Algorithmic bytes / flops per unit", summed over `layers` = (M, K, N, r, has_bias, dx_ran, reads_context)
    per LoRA linear (reference operator lora_diffusion/lora.py:49-50 and its autograd): forward e·(MK+NK+MN) + e·r(K+N)
    (+ e·N bias), backward e·(MN+MK) + (e+4)·r(K+N) + [e·(NK+MK) when dX is needed]; flops 2MKN + 2Mr(K+N) forward,
    [2MKN] + 4Mr(K+N) + [2Mr(K+N)] backward.  Every operand once per direction, whatever the kernels re-read.
    contract: dX needed unless the layer reads the (frozen) encoder context and the step produced none — SURVEY's rule."""
    e = float(esize)
    fb = bb = ff = bf = 0.0
    n = 0
    for M, K, N, r, has_bias, dx_ran, reads_context in layers:
        dx = (dx_ran or not reads_context) if contract else dx_ran
        fb += e * (M * K + N * K + M * N) + e * r * (K + N) + (e * N if has_bias else 0.0)
        bb += e * (M * N + M * K) + (e + 4.0) * r * (K + N) + (e * (N * K + M * K) if dx else 0.0)
        ff += 2.0 * M * K * N + 2.0 * M * r * (K + N)
        bf += (2.0 * M * K * N if dx else 0.0) + 4.0 * M * r * (K + N) + (2.0 * M * r * (K + N) if dx else 0.0)
        n += 1
    return {"layers": n, "fwd_bytes": fb, "bwd_bytes": bb, "fwd_flops": ff, "bwd_flops": bf}


def lr_lambda(name: str, num_warmup_steps: int = 0, num_training_steps: Optional[int] = None, lr_init: float = 1.0):
    """λ(epoch): the factor the trainers' `get_scheduler(name, optimizer, num_warmup_steps, num_training_steps)` puts on every
    param group's learning rate after `epoch` calls of `lr_scheduler.step()` (train_lora_dreambooth.py:737-743 with `--lr_scheduler`
    choices :345-353, default "constant"; cli_lora_pti.py:746-751, default "linear" with 0 warm-up steps :534-535).  The function
    is diffusers' (`diffusers.optimization`, a torch LambdaLR per name) and not part of the reference tree: the formulas are
    restated from its published definitions — parity unpinned for them; the LambdaLR mechanics around them (λ(0) at
    construction, one increment per step(), lr = base_lr·λ) are torch's and are what the fixture pins."""
    w = int(num_warmup_steps)
    if name not in SCHEDULER_NAMES:
        raise ValueError(f"unknown lr_scheduler {name!r}; one of {SCHEDULER_NAMES}")
    if name == "constant":
        return lambda e: 1.0
    if name == "constant_with_warmup":
        return lambda e: float(e) / float(max(1.0, w)) if e < w else 1.0
    if num_training_steps is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")
    n = int(num_training_steps)
    ramp = lambda e: float(e) / float(max(1, w))
    if name == "linear":
        return lambda e: ramp(e) if e < w else max(0.0, float(n - e) / float(max(1, n - w)))
    if name == "cosine":
        return lambda e: ramp(e) if e < w else max(
            0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * (float(e - w) / float(max(1, n - w))))))
    if name == "cosine_with_restarts":
        def f(e):
            if e < w:
                return ramp(e)
            progress = float(e - w) / float(max(1, n - w))
            return 0.0 if progress >= 1.0 else max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((1.0 * progress) % 1.0))))
        return f
    lr_end, power = 1e-7, 1.0  # "polynomial"

    def poly(e):
        if e < w:
            return ramp(e)
        if e > n:
            return lr_end / lr_init
        remaining = 1 - (e - w) / (n - w)
        return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init
    return poly


def lora_layers(model: nn.Module) -> List[LoraInjectedLinear]:
    """All LoraInjectedLinear modules in `model.modules()` order — for an injected model this is the
    enumeration order of lora.py:78-114, because every target is visited under its first matching ancestor."""
    return [m for m in model.modules() if isinstance(m, LoraInjectedLinear)]


class LayerSink:
    """Where one layer's factor gradients go: device pointers of row block 0 of its `up` (gB) and `down` (gA) slots in the
    slab's partial-sum buffer.  The backward of a layer does not launch anything for them: it DEFERS two problems to the
    slab, which launches all problems of a pass together (csrc/lora_grad.hip: lora_grad_batched)."""

    def __init__(self, slab, index: int, up_off: int, down_off: int):
        self.slab, self.index = slab, index
        base = slab.partials.data_ptr()
        self.up_ptr, self.down_ptr = base + 4 * up_off, base + 4 * down_off

    def defer_layer(self, dy2, x2, t, u, scale: float, need_dx: bool = True):
        M, N = dy2.shape
        K, r = x2.shape[1], t.shape[1]
        stride = self.slab.stride
        self.slab.note_layer(self.index, M, need_dx)
        self.slab.defer(nat.grad_problem(dy2, 0, N, N, t, 0, r, r, [self.up_ptr], r, False, stride, M, scale), self.index,
                        (dy2, t))
        self.slab.defer(nat.grad_problem(x2, 0, K, K, u, 0, r, r, [self.down_ptr], r, True, stride, M, scale), None,
                        (x2, u))


class LoraSlab:
    """Re-homes every LoRA factor of `models` into one flat fp32 parameter slab (+ gradient slab).

    The nn.Parameter objects stay the same (optimizers, generators and state_dict keep working); only their
    storage moves; `.grad` of each Parameter is a view of the gradient slab.  Each layer gets a `_dfa_grad_sink`
    (LayerSink): its backward defers its two factor-gradient problems here, `flush()` launches everything deferred so
    far in a few chip-filling launches (the operands — every layer's dY and X — simply stay alive until then: 288 GB of
    HBM) and folds the row-block partial sums of the layers that ran into the gradient slab, in block order."""

    def __init__(self, models: Sequence[nn.Module], dense_params: Sequence[nn.Parameter] = ()):
        """dense_params: trainable non-LoRA Parameters that ride in the same flat buffers BEHIND the LoRA region (the token
        embedding table of a text encoder whose input embeddings train, cli_lora_pti.py:706-722): same clip norm, same
        fused AdamW, their own learning-rate range — but no partial-sum slab and no place in the LoRA gradient exchange."""
        self._sinks = []
        self.layers: List[LoraInjectedLinear] = []
        self.model_ranges: List[Tuple[int, int]] = []
        self.models = list(models)
        for model in models:
            start = sum(l.lora_up.weight.numel() + l.lora_down.weight.numel() for l in self.layers)
            self.layers += lora_layers(model)
            end = sum(l.lora_up.weight.numel() + l.lora_down.weight.numel() for l in self.layers)
            self.model_ranges.append((start, end))
        if not self.layers:
            raise ValueError("No lora injected.")
        from .attention import _attach_dropin_groups

        for model in models:  # groups the attention switch built for a slab-less trainer: this slab brings its own (or none)
            _attach_dropin_groups(model, False)
        device = self.layers[0].lora_up.weight.device
        if device.type != "cuda":
            raise RuntimeError("LoraSlab: the model must be on the HIP device")
        total = sum(l.lora_up.weight.numel() + l.lora_down.weight.numel() for l in self.layers)
        pad = (-total) % 4  # keep 16-byte granularity for vector loads
        self.numel = total
        self.stride = total + pad
        self.device = device
        self.dense_ranges: List[Tuple[int, int]] = []
        tail = 0
        for q in dense_params:
            self.dense_ranges.append((self.stride + tail, self.stride + tail + q.numel()))
            tail += (q.numel() + 3) // 4 * 4
        self.total = self.stride + tail  # LoRA region (padded) + dense tail: what the clip norm and the optimizer cover
        self.params = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.total, dtype=torch.float32, device=device)
        for q, (a, b) in zip(dense_params, self.dense_ranges):
            view = self.params[a:b].view(q.shape)
            view.copy_(q.detach().float())
            q.data = view
            q.grad = self.grads[a:b].view(q.shape)
        # row-block partial sums of the factor gradients, [blocks][slab]: written by the gradient kernel with plain
        # stores, summed in block order by the fold (deterministic, no atomics).  Never needs zeroing: a layer's blocks
        # are fully rewritten whenever it runs, and only layers that ran are folded.
        self.partials = torch.empty((nat.GRAD_MAX_BLOCKS, total + pad), dtype=torch.float32, device=device)
        self.offsets = []
        self._pending, self._keep, self._ran = {}, [], {}
        self._range_tables = {}
        # factor gradients of a pass in one launch (lora_grad_planned); off: <= 28 problems per launch (DFA_ONE_LAUNCH_GRADS=0: A/B knob)
        self.one_launch_grads = os.environ.get("DFA_ONE_LAUNCH_GRADS", "1") != "0"
        self._plan_need, self._recording_plans, self._recorded_plans = {}, {}, []
        self.layer_rows = {}  # layer index -> (rows M, dX produced) of its last backward (accounting: survey_work)
        # layers whose input is the text encoder's output (diffusers names the cross-attention of a transformer block attn2):
        # the ones SURVEY §8(a) a3 exempts from dX while the encoder is frozen
        name_of = {id(m): n for model in self.models for n, m in model.named_modules()}
        self.reads_context = [name_of.get(id(l), "").endswith(("attn2.to_k", "attn2.to_v")) for l in self.layers]
        self.qkv_groups, self.ctx_groups = [], []
        self.packed = None
        off = 0
        for index, layer in enumerate(self.layers):
            for attr in ("lora_up", "lora_down"):  # file order: up then down
                p = getattr(layer, attr).weight
                n = p.numel()
                pv = self.params[off:off + n].view(p.shape)
                pv.copy_(p.detach().float())
                p.data = pv
                p.grad = self.grads[off:off + n].view(p.shape)
                self.offsets.append((off, n))
                off += n
            sink = LayerSink(self, index, self.offsets[-2][0], self.offsets[-1][0])
            self._sinks.append(sink)
            layer.__dict__["_dfa_grad_sink"] = sink

    # -- grouped projections ---------------------------------------------------------------------
    def enable_groups(self, split_ctx_at: Sequence[str] = ()):
        """(split_ctx_at: name prefixes — e.g. ("down_blocks.",) — whose cross-attentions get a K/V group of their OWN instead
        of joining the model-wide one: the gradients of the other group are then final as soon as backward has left ITS
        blocks, which is what an early exchange bucket needs — LoraTrainer(early_bucket=True).)
        Finds the attention modules under the slab's models and groups their LoRA projections (groups.py): to_q/to_k/
        to_v of a self-attention into a QKVGroup, to_k/to_v of all cross-attentions over the same context width into a
        CtxKVGroup.  The groups are used by the attention forward installed with the reference's attention switch
        (attention.py); a model whose attention runs through its own forward never looks at them."""
        from .attention import _is_attention_module
        from .groups import CtxKVGroup, QKVGroup

        index_of = {id(l): i for i, l in enumerate(self.layers)}
        for model in self.models:
            cross = {}
            for name, m in model.named_modules():
                if not _is_attention_module(m):
                    continue
                trio = [m.to_q, m.to_k, m.to_v]
                if not all(id(l) in index_of for l in trio):
                    continue
                # cross-attention: keys/values come from another width, or the block calls it `attn2` (the name diffusers'
                # BasicTransformerBlock gives its cross-attention in every version the reference supports)
                is_cross = (m.to_k.linear.in_features != m.to_q.linear.in_features) or name.split(".")[-1] == "attn2"
                if is_cross:
                    key = (m.to_k.linear.in_features, m.to_k.lora_down.weight.shape[0],
                           next((pre for pre in split_ctx_at if name.startswith(pre)), ""))
                    cross.setdefault(key, []).append(m)
                elif QKVGroup.eligible(trio):
                    grp = QKVGroup(trio, [self._sinks[index_of[id(l)]] for l in trio])
                    m.__dict__["_dfa_qkv"] = grp
                    self.qkv_groups.append(grp)
            # transformers' CLIPAttention (LoRA target class of the text encoder, lora.py:54): q_proj / k_proj / v_proj multiply
            # the same hidden states and are called one after the other by a forward this package does not replace — the
            # members share one launch through groups.shared_projection (biases ride along; 3·r ≤ 16 rank slots)
            for name, m in model.named_modules():
                if m.__class__.__name__ != "CLIPAttention" or not all(hasattr(m, a) for a in ("q_proj", "k_proj", "v_proj")):
                    continue
                trio = [m.q_proj, m.k_proj, m.v_proj]
                if all(id(l) in index_of for l in trio) and QKVGroup.eligible(trio):
                    grp = QKVGroup(trio, [self._sinks[index_of[id(l)]] for l in trio])
                    for i, l in enumerate(trio):
                        l.__dict__["_dfa_shared"] = (grp, i)
                    self.qkv_groups.append(grp)
            for mods in cross.values():
                layers = [l for m in mods for l in (m.to_k, m.to_v)]
                if len(mods) < 2 or not CtxKVGroup.eligible(layers):
                    continue
                grp = CtxKVGroup(mods, layers, [self._sinks[index_of[id(l)]] for l in layers])
                for i, m in enumerate(mods):
                    m.__dict__["_dfa_ctx"] = (grp, i)
                self.ctx_groups.append(grp)
                model.register_forward_pre_hook(lambda module, args, g=grp: g.new_pass())

    # -- packed compute-dtype factors ------------------------------------------------------------
    def enable_packed(self, dtype: torch.dtype):
        """Allocates the packed-factor buffer — per layer Apack (32·K) and Bpack (32·N) in the compute dtype, both
        orientations (lora_hip.h), plus the operands of the grouped projections — and the device table for the one-launch
        re-pack (lora_pack_items: one row per factor and destination)."""
        from .groups import bind_ctx_views, bind_qkv_views, ctx_pack_rows, qkv_pack_rows

        rows, off = [], 0
        layer_views = []
        for i, layer in enumerate(self.layers):
            r, K = layer.lora_down.weight.shape
            N = layer.lora_up.weight.shape[0]
            if r > 16:
                continue
            up_off, down_off = self.offsets[2 * i][0], self.offsets[2 * i + 1][0]
            # {src_off, which, len, r, d16_off, d16_ld, dT_off, rows}
            rows.append([down_off, 0, K, r, off, K, off + 16 * K, 16])                       # A16 [16,K] | At16 [K,16]
            rows.append([up_off, 1, N, r, off + 32 * K, N, off + 32 * K + 16 * N, 16])       # Bt16 [16,N] | B16 [N,16]
            layer_views.append((layer, off, K, N))
            off += 32 * (K + N)
        index_of = {id(l): i for i, l in enumerate(self.layers)}
        group_views = []

        def src_offs(grp):  # (up_off, down_off) of every member's fp32 factors inside the slab
            return [(self.offsets[2 * index_of[id(l)]][0], self.offsets[2 * index_of[id(l)] + 1][0]) for l in grp.layers]

        for grp in self.qkv_groups:
            g_rows, spec, used = qkv_pack_rows(grp, src_offs(grp), off)
            rows += g_rows
            group_views.append((grp, spec, bind_qkv_views))
            off += used
        for grp in self.ctx_groups:
            g_rows, spec, used = ctx_pack_rows(grp, src_offs(grp), off)
            rows += g_rows
            group_views.append((grp, spec, bind_ctx_views))
            off += used
        if not rows:
            self.packed = None
            return
        self.packed = torch.zeros(off, dtype=dtype, device=self.device)  # zeroed ONCE: block-diagonal groups rely on it
        self._pack_table = torch.tensor(rows, dtype=torch.int64).to(self.device)
        self._pack_maxlen = max(r_[2] for r_ in rows)
        pk = self.packed
        for layer, o, K, N in layer_views:
            layer.__dict__["_dfa_packed"] = (pk[o:o + 32 * K], pk[o + 32 * K:o + 32 * (K + N)])
        for grp in self.qkv_groups:  # the members' own Bt16 [16,N] tiles (P-only launches of a wide group without dX)
            grp.Fb_part = [l.__dict__["_dfa_packed"][1][:16 * grp.N] for l in grp.layers]
        for grp, spec, bind in group_views:
            bind(grp, pk, spec)
        self.repack()

    def repack(self):
        """Refresh every packed factor from the fp32 master slab (one launch)."""
        if self.packed is not None:
            nat.lora_pack_items(self._pack_table, self._pack_table.shape[0], self._pack_maxlen, self.params, self.packed)

    # -- accounting ------------------------------------------------------------------------------
    def note_layer(self, index: int, rows: int, need_dx: bool):
        self.layer_rows[index] = (int(rows), bool(need_dx))

    def survey_work(self, esize: int = 2, contract: bool = False):
        """Algorithmic bytes / flops of the LoRA layers that ran in the last step, by SURVEY §8(d)'s per-layer formulas
        (`survey_bytes_flops`).  contract=False: a dX is counted where the step produced one; contract=True: SURVEY's own
        accounting — a dX for every layer but the `attn2.to_k / to_v` of a frozen text encoder (§8(a) a3), which is how its
        5 331 MB for config 2 come about (the step itself skips three more: the first block's q/k/v read a tensor nothing
        trainable precedes — 32 MB less)."""
        rows = []
        for i, (M, dx) in self.layer_rows.items():
            layer = self.layers[i]
            r, K = layer.lora_down.weight.shape
            N = layer.lora_up.weight.shape[0]
            rows.append((M, K, N, r, layer.linear.bias is not None, dx, self.reads_context[i]))
        return survey_bytes_flops(rows, esize, contract)

    # -- factor gradients --------------------------------------------------------------------------
    def zero_grad(self):
        self.grads.zero_()
        self._pending, self._keep, self._ran = {}, [], {}

    def defer(self, problem, layer_index, keep):
        """Queue one factor-gradient problem (a `_native.GradProblem`); `keep` = the tensors it reads, held until the
        launch; `layer_index` marks the layer whose slab range must be folded afterwards (None: covered by a sibling)."""
        if layer_index is not None and layer_index in self._ran:
            # the layer ran backward once more before a flush (gradient accumulation over micro-batches, a shared module):
            # its partial-sum slots hold one pass, so launch and fold what is pending first — the fold accumulates
            self.flush()
        dt = keep[0].dtype
        self._pending.setdefault(dt, []).append(problem)
        self._keep.append(keep)
        if layer_index is not None:
            self._ran[layer_index] = nat.grad_row_blocks(problem.M)

    def flush(self):
        """Launch every deferred problem, then fold the partial sums of the layers that ran into `grads` (+=)."""
        for grp in self.ctx_groups:
            grp.check_pass_complete()
        if not self._pending:
            return
        for dt, problems in self._pending.items():
            if not self._launch_planned(problems, dt):
                nat.lora_grad_batched(problems, dt, self.device)
        key = tuple(sorted(self._ran.items()))
        table = self._range_tables.get(key)
        if table is None:
            rows = [[self.offsets[2 * i][0], self.offsets[2 * i][1] + self.offsets[2 * i + 1][1], nb, 0] for i, nb in key]
            table = self._range_tables[key] = (torch.tensor(rows, dtype=torch.int64).to(self.device),
                                               max(r_[1] for r_ in rows))
        nat.lora_fold_partials(table[0], len(key), table[1], self.partials, self.stride, self.grads, True)
        self._pending, self._keep, self._ran = {}, [], {}

    # -- all factor-gradient problems of a pass in ONE launch (a plan in device memory) -----------------------------
    def prepare_recording(self):
        """Call right before a step is recorded into a hipGraph (outside the capture): reserves the pinned host buffers the
        recording's plan copies will read on EVERY replay.  They belong to that recording (`take_recording_plans`), never to
        the slab: a host-launched step or a later recording must not be able to rewrite what an older recording replays."""
        self._recording_plans = {dt: torch.empty(2 * need + 4096, dtype=torch.uint8, pin_memory=True)
                                 for dt, need in self._plan_need.items()}
        self._recorded_plans = []

    def take_recording_plans(self):
        plans, self._recorded_plans, self._recording_plans = self._recorded_plans, [], {}
        return plans

    def _launch_planned(self, problems, dt) -> bool:
        if not self.one_launch_grads:
            return False
        host = None
        if torch.cuda.is_current_stream_capturing():
            host = self._recording_plans.pop(dt, None)  # one buffer per (recording, dtype): a second flush of the pass declines
            if host is None:
                return False
        else:
            self._plan_need[dt] = max(self._plan_need.get(dt, 0), nat.lora_grad_plan_bytes(problems))
        plan = nat.lora_grad_one_launch(problems, dt, self.device, host)
        if plan is None:
            return False
        if host is not None:
            self._recorded_plans.append(plan)  # the recording's: its copy node reads plan[0] on every replay
        else:
            # host-launched: `_keep` is cleared at the end of this very flush, so what keeps the plan's buffers alive until the
            # copy and the launch have run is the allocators' stream ordering (the device buffer is reused only by later work of
            # this stream; the pinned allocator holds a block until the copy that reads it has completed)
            self._keep.append(plan)
        return True

    def detach_sinks(self):
        for layer in self.layers:
            layer.__dict__.pop("_dfa_grad_sink", None)
            layer.__dict__.pop("_dfa_packed", None)
            layer.__dict__.pop("_dfa_shared", None)
        for model in self.models:
            for m in model.modules():
                m.__dict__.pop("_dfa_qkv", None)
                m.__dict__.pop("_dfa_ctx", None)

    def range_of(self, module: nn.Module) -> Tuple[int, int]:
        """[start, end) of the slab covering the LoRA layers under `module` (must be contiguous)."""
        inside = {id(l) for l in lora_layers(module)}
        idx = [i for i, l in enumerate(self.layers) if id(l) in inside]
        if not idx:
            return (0, 0)
        assert idx == list(range(idx[0], idx[-1] + 1)), "layers under the module are not contiguous in the slab"
        return (self.offsets[2 * idx[0]][0], self.offsets[2 * idx[-1] + 1][0] + self.offsets[2 * idx[-1] + 1][1])


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

    def step(self, grad_mul: float = 1.0, lr_mul: float = 1.0):
        """One optimizer step on the current gradient slab; `grad_mul` = 1/(world_size·loss_scale); `lr_mul` = the learning-rate
        schedule's factor for this step (torch LambdaLR: every group's lr = its base lr · λ, a host scalar per launch).  `step_count` counts
        calls (it keys the on-device noise draw, which the reference also advances every iteration); the bias
        corrections use the number of APPLIED steps, kept on the device (`norm[2]`): an overflowed step is skipped and
        does not count, like torch.cuda.amp.GradScaler never calls optimizer.step() for it."""
        s = self.slab
        self.step_count += 1
        # (one norm over everything that trains: clip_grad_norm_ over chain(unet.parameters(), text_encoder.parameters()),
        #  cli_lora_pti.py:448-450; the padding between the LoRA region and a dense tail is zero)
        nat.lora_grad_sqnorm(s.grads[: s.numel] if s.total == s.stride else s.grads, grad_mul, self.norm)
        for g in self.groups:
            a, b = g["range"]
            if b <= a:
                continue
            if g.get("rows") is not None:  # a table of which few rows ever get a gradient (TokenTable): same result, less traffic
                V, D, active = g["rows"]
                nat.lora_adamw_rows(s.params[a:b].view(V, D), s.grads[a:b].view(V, D), self.exp_avg[a:b].view(V, D),
                                    self.exp_avg_sq[a:b].view(V, D), active, self.norm, grad_mul, self.max_grad_norm, g["lr"] * lr_mul,
                                    self.betas[0], self.betas[1], self.eps, g.get("weight_decay", 1e-2), 0)
                continue
            nat.lora_adamw_step(s.params[a:b], s.grads[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], self.norm,
                                grad_mul, self.max_grad_norm, g["lr"] * lr_mul, self.betas[0], self.betas[1], self.eps,
                                g.get("weight_decay", 1e-2), 0)

    def grad_norm(self) -> float:
        """Total (pre-clip) gradient L2 norm of the last step (host sync)."""
        return math.sqrt(float(self.norm[0].item()))

    def overflowed(self) -> bool:
        return bool(self.norm[1].item() != 0.0)

    def applied_steps(self) -> int:
        """Optimizer steps that really updated the parameters (host sync)."""
        return int(self.norm[2].item())

    def skipped_steps(self) -> int:
        return int(self.norm[3].item())


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
        # single: the whole slab always goes in ONE all-reduce after backward, whatever early_range says.  Set by the
        # trainer whenever a recorded (hipGraph) step was ASKED for: a replayed step can only exchange that way, and a rank
        # whose recording failed — or whose step is not recordable — must keep issuing the same collectives as its peers.
        self.single = False
        self._armed = False
        self._early_sent = False
        self._pending = []

    def arm(self):
        """Call before each backward pass; launch_early() then fires at most once."""
        self._armed = self.active and self.early_range is not None and not self.single
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


class LossScaler:
    """torch.cuda.amp.GradScaler's scale schedule at a FIXED LAG, so that every data-parallel rank changes its scale at
    the same step: an overflowed (hence skipped) step halves the loss scale, `growth_interval` clean steps double it again
    (never above the initial value).  The overflow flag of step k is shipped to the host without a wait when step k ends
    (`watch`) and is applied at the start of step k + LAG (`begin_step`), after a wait on an event that completed long
    ago — never "whenever the copy happens to have landed", which differs from rank to rank: ranks that disagree on the
    scale for even one step mix gradients scaled by S and S/2 in the SUM all-reduce and drift apart for good.  The flag
    itself is computed on the all-reduced slab, so it is identical everywhere.  Device-agnostic (`watch` takes a callable
    that returns the flag once awaited): the GPU trainer hands it a pinned-buffer reader, the CPU tests plain floats."""

    LAG = 2

    def __init__(self, initial: float, growth_interval: int = 2000):
        self.scale = float(initial)
        self.initial = float(initial)
        self.growth_interval = int(growth_interval)
        self.clean_steps = 0
        self._inflight = []  # FIFO of callables, one per finished step

    def watch(self, read_flag):
        self._inflight.append(read_flag)

    def begin_step(self) -> bool:
        """Applies the decision of the step that ended LAG steps ago; True when the scale changed."""
        if len(self._inflight) < self.LAG:
            return False
        overflow = float(self._inflight.pop(0)()) != 0.0
        before = self.scale
        if overflow:
            self.scale = max(1.0, self.scale * 0.5)
            self.clean_steps = 0
        else:
            self.clean_steps += 1
            if self.clean_steps >= self.growth_interval and self.scale < self.initial:
                self.scale, self.clean_steps = min(self.initial, self.scale * 2.0), 0
        return self.scale != before


class _TokenRowsFn(torch.autograd.Function):
    """rows = table[ids] (nn.Embedding.forward of the text encoder's token table) through the HIP gather; backward hands the
    incoming gradient rows to the TokenTable, which sums them per token in a fixed order after the exchange — the table's
    `.grad` is never produced by autograd."""

    @staticmethod
    def forward(ctx, ids, weight, table, out_dtype):
        ctx.table = table
        ctx.ids = ids
        return nat.embed_rows_fwd(weight.detach(), ids, out_dtype)

    @staticmethod
    def backward(ctx, d_rows):
        t = ctx.table
        rows = d_rows.reshape(-1, t.D)
        t._pending.append((ctx.ids.reshape(-1).contiguous(), rows if rows.is_contiguous() else rows.contiguous()))
        return None, None, None, None


class TokenTable:
    """The input-embedding table of a text encoder whose token embeddings train next to the LoRA factors: the tuning phase
    of lora_diffusion/cli_lora_pti.py with continue_inversion (default, :528) — `text_encoder.get_input_embeddings()
    .parameters()` in the optimizer (:706-722), everything else of the encoder frozen.  The fp32 master lives in the slab's
    dense tail (clip + AdamW like every other trainable); the module's forward is the HIP gather; the gradient rows of the
    tokens that occurred are summed per token in position order (`embed_rows_bwd`) — after an all-gather of (ids, rows) in
    rank order under data parallelism, so every replica adds the same numbers in the same order and only B·L rows (not the
    200-MB table gradient) cross the GPUs."""

    def __init__(self, slab: "LoraSlab", text_encoder: nn.Module, index: int, out_dtype: torch.dtype):
        import functools

        self.module = text_encoder.get_input_embeddings()
        w = self.module.weight
        self.V, self.D = w.shape
        self.range = slab.dense_ranges[index]
        a, b = self.range
        self.grad = slab.grads[a:b].view(self.V, self.D)
        self.out_dtype = out_dtype
        self.active = torch.zeros(self.V, dtype=torch.uint8, device=w.device)  # rows that ever had a gradient (lora_adamw_rows)
        self._pending = []
        self.module.forward = functools.partial(self._forward, self.module)  # (instance attribute: the class is untouched)

    def check_ids(self, input_ids):
        """torch.nn.Embedding raises on an id outside the table; the HIP gather cannot (it poisons the row with NaN, so the
        step is skipped as an overflow).  Raise like torch where the check is FREE: ids still on the host.  Device-resident ids
        are not looked at (the check was two blocking device-to-host syncs in every host-launched encoder pass, and
        under data parallelism it raised on one rank only, leaving the others blocked in the next collective): for them the
        NaN-poisoned row is the signal — the step is skipped like any overflow, on every rank alike, since the overflow flag is
        computed on the all-reduced slab."""
        if input_ids.device.type != "cpu" or not input_ids.numel():
            return
        if int(input_ids.min()) < 0 or int(input_ids.max()) >= self.V:
            raise IndexError(f"token id out of range for the {self.V}-row embedding table")

    def _forward(self, module, input_ids):
        self.check_ids(input_ids)
        dt = torch.get_autocast_dtype("cuda") if (torch.is_autocast_enabled("cuda") and self.out_dtype != torch.float32) \
            else self.out_dtype
        if not module.weight.requires_grad or not torch.is_grad_enabled():
            return nat.embed_rows_fwd(module.weight.detach(), input_ids, dt)
        return _TokenRowsFn.apply(input_ids, module.weight, self, dt)

    def begin_pass(self):
        self._pending = []

    def collect(self, pg, world: int):
        """Table gradient of the passes since begin_pass(): rows of the tokens that occurred, summed in position order."""
        for k, (ids, rows) in enumerate(self._pending):
            if world > 1:
                all_ids = [torch.empty_like(ids) for _ in range(world)]
                all_rows = [torch.empty_like(rows) for _ in range(world)]
                dist.all_gather(all_ids, ids, group=pg)
                dist.all_gather(all_rows, rows, group=pg)
                ids, rows = torch.cat(all_ids), torch.cat(all_rows)
            nat.embed_rows_bwd(rows, ids, self.grad, accumulate=k > 0, active=self.active)


class LoraTrainer:
    """One object per process (= per GPU).  `step()` runs one full training step and returns the loss tensor
    (no host sync unless the caller reads it)."""

    def __init__(self, unet: nn.Module, text_encoder: Optional[nn.Module] = None, lr=1e-4, lr_text=5e-6,
                 weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, loss_scale: Optional[float] = None,
                 v_prediction=False, process_group=None, always_reduce=False, capture_graph=False,
                 group_projections=True, lr_embed: float = 5e-4, weight_decay_embed: Optional[float] = None,
                 lr_scheduler: str = "constant", lr_warmup_steps: int = 0, max_train_steps: Optional[int] = None,
                 scheduler_steps_first: bool = False, early_bucket: bool = False, scheduler_steps_per_call: int = 1):
        """capture_graph: record add_noise → [text encoder] → UNet forward → loss → backward → factor gradients of a step
        once into a hipGraph and replay it on later steps with the same shapes (inputs are copied into static buffers).
        The gradient exchange and the optimizer stay outside the graph, so no collective is ever captured.  A step with a
        LoRA text encoder is recordable when the caller hands over `input_ids` (the encoder's forward then belongs to the
        step, train_lora_dreambooth.py:840); a mask is a static input like the latents.  Anything else runs host-launched.
        group_projections: run attn1 to_q/to_k/to_v as one launch per block and the attn2 to_k/to_v of all blocks as one
        launch per pass (groups.py; effective for attention modules switched to the HIP cores with the reference's
        `set_use_memory_efficient_attention_xformers`).
        Token embeddings: when `text_encoder.get_input_embeddings().weight.requires_grad` is set — what the PTI tuning phase
        does under continue_inversion (cli_lora_pti.py:706-722) — the table trains too, with `lr_embed` (continue_inversion_lr /
        learning_rate_ti, :713-715) and `weight_decay_embed` (default: `weight_decay`, one AdamW for all groups, :738).
        Learning-rate schedule: `lr_scheduler` / `lr_warmup_steps` / `max_train_steps` are the trainers' arguments of the same
        names (train_lora_dreambooth.py:298,345-358,737-743; cli_lora_pti.py: `lr_scheduler_lora`, `lr_warmup_steps_lora`,
        `max_train_steps_tuning`, :534-535,746-751) and give λ (`lr_lambda`) on every group's lr.  `scheduler_steps_first`:
        the dreambooth loop steps the scheduler AFTER the optimizer (:885-886: step k runs at λ(k), k = 0, 1, …), the PTI
        tuning loop BEFORE it (cli_lora_pti.py:434: step k runs at λ(k + 1) — with the default linear schedule the first step
        is already at lr·(1 − 1/N) and the last at 0).  The factor is a host scalar of the AdamW launch, which is outside a
        recorded step anyway.
        `scheduler_steps_per_call`: how far ONE `lr_scheduler.step()` of the trainer moves the schedule.  The
        dreambooth script hands its scheduler to `accelerator.prepare` (train_lora_dreambooth.py:750-757), and accelerate's
        AcceleratedScheduler steps the wrapped scheduler once PER PROCESS on every call (split_batches=False, the script's
        default): on N GPUs the reference's non-constant schedules decay N times faster per optimizer step than on one.  Pass the
        world size to reproduce that route; 1 (default) is a plain scheduler — cli_lora_pti.py (single process, no accelerate)
        and the default "constant" schedule are unaffected either way.  accelerate is not in the reference tree nor importable
        here: that N× rule is restated from its published AcceleratedScheduler, parity unpinned.  Not reproduced: accelerate
        also holds the scheduler back on a step its GradScaler skipped — the skip is known here only two steps later
        (LossScaler's fixed lag), so the schedule counts every step() call; it matters for none of the reference's fp16
        defaults ("constant"; PTI has no scaler)."""
        self.unet, self.text_encoder = unet, text_encoder
        self.early_bucket = bool(early_bucket)
        self.lr_lambda = lr_lambda(lr_scheduler, lr_warmup_steps, max_train_steps, lr_init=lr)
        self.scheduler_steps_first = bool(scheduler_steps_first)
        self.scheduler_epoch = 0  # LambdaLR.last_epoch: scheduler steps taken so far
        self.scheduler_steps_per_call = int(scheduler_steps_per_call)
        if self.scheduler_steps_per_call < 1:
            raise ValueError("scheduler_steps_per_call must be >= 1")
        self.capture_graph = bool(capture_graph)
        self._graph = None
        models = [unet] + ([text_encoder] if text_encoder is not None and lora_layers(text_encoder) else [])
        emb = text_encoder.get_input_embeddings() if (text_encoder is not None and hasattr(text_encoder, "get_input_embeddings")) \
            else None
        train_emb = emb is not None and emb.weight.requires_grad
        self.trains_text_encoder = len(models) > 1 or train_emb
        self.slab = LoraSlab(models, [emb.weight] if train_emb else [])
        if group_projections:
            # early_bucket: the cross-attentions of the down blocks project their K/V in a launch of their own, so that the
            # [up|mid] part of the slab is final when backward leaves the mid block (the bucket hook below) — the price of
            # an early exchange bucket WITH grouped projections: one more projection launch per pass, one more P-only
            # launch in backward
            self.slab.enable_groups(split_ctx_at=("down_blocks.",) if self.early_bucket else ())
        groups = [{"range": self.slab.model_ranges[0], "lr": lr, "weight_decay": weight_decay}]
        if len(models) > 1:
            groups.append({"range": self.slab.model_ranges[1], "lr": lr_text, "weight_decay": weight_decay})
        if train_emb:
            groups.append({"range": self.slab.dense_ranges[0], "lr": lr_embed,
                           "weight_decay": weight_decay if weight_decay_embed is None else weight_decay_embed})
        self.opt = FusedClipAdamW(self.slab, groups, betas, eps, max_grad_norm)
        self.device = self.slab.params.device
        self.dtype = next(p for p in unet.parameters() if p.dim() == 4).dtype  # conv weight dtype = compute dtype
        self.tail_events = None  # a list switches on the timing of the host-launched tail of every replayed step
        self.token_table = None
        if train_emb:
            te_dtype = next((p.dtype for n_, p in text_encoder.named_parameters() if p is not emb.weight), torch.float32)
            self.token_table = TokenTable(self.slab, text_encoder, 0, te_dtype)
            self.opt.groups[-1]["rows"] = (self.token_table.V, self.token_table.D, self.token_table.active)
        initial = float(loss_scale) if loss_scale is not None else (1024.0 if self.dtype == torch.float16 else 1.0)
        self.scaler = LossScaler(initial, self.GROWTH_INTERVAL)
        self._warned_overflow = False
        # one pinned word + one event per step in flight (LAG + 1: the slot of step k is reused at step k + LAG + 1)
        self._flag_slots = ([(torch.zeros(1, dtype=torch.float32).pin_memory(), torch.cuda.Event())
                             for _ in range(LossScaler.LAG + 1)] if initial != 1.0 else None)
        self._flag_turn = 0
        self.slab.enable_packed(self.dtype)
        self.v_prediction = v_prediction
        self.sqrt_acp, self.sqrt_1macp = ddpm_tables(device=self.device)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.exchange = SlabExchange(self.slab.grads, self.slab.numel, process_group, always_reduce=always_reduce)
        # A recorded step exchanges the slab in one all-reduce.  Once recording was ASKED for, every step of this trainer
        # does — also the host-launched ones (recording failed on this rank only, a step that is not recordable): the
        # collective sequence then depends on what was requested, which is the same on every rank, never on what succeeded.
        self.exchange.single = self.capture_graph
        if self.exchange.active:
            self._broadcast_initial_state()
            self._install_bucket_hook()

    @property
    def loss_scale(self) -> float:
        return self.scaler.scale

    @loss_scale.setter
    def loss_scale(self, value: float):
        self.scaler.scale = float(value)

    # -- data parallel -------------------------------------------------------------------------
    def _broadcast_initial_state(self):
        """DDP construction broadcasts module state from rank 0 (train_lora_dreambooth.py:751-757); the frozen
        base is loaded identically everywhere, so only the LoRA slab travels."""
        dist.broadcast(self.slab.params, src=0, group=self.pg)

    def _install_bucket_hook(self):
        """Two buckets in backward-completion order.  Enumeration order is down, up, mid; backward finishes the
        up blocks first, then mid, then the down blocks — so [up|mid] is a contiguous early bucket whose
        all-reduce overlaps the rest of the backward pass (its factor gradients are launched and folded first).
        Not with a CtxKVGroup: the to_k/to_v gradients of EVERY block then come out of one launch at the end of
        backward, so no slab range is final before that — the whole slab (5 MB at rank 4) goes in one all-reduce.
        (Restoring the overlap there means cutting the K/V group by block range — one group for [up|mid], one for
        down — at the price of a second 13-µs projection launch per pass; the exchange it would hide is ≈ 60–100 µs of a
        30-ms step.  DESIGN.md §6.)"""
        mid = getattr(self.unet, "mid_block", None)
        ups = getattr(self.unet, "up_blocks", None)
        if mid is None or ups is None:
            return
        if self.slab.ctx_groups and not self.early_bucket:
            return  # one model-wide K/V group: its gradients come out of one launch at the very end of backward
        inside = {id(l) for l in lora_layers(ups)} | {id(l) for l in lora_layers(mid)}
        for grp in self.slab.ctx_groups:  # a group must lie entirely inside or entirely outside the early bucket
            member = [id(l) in inside for l in grp.layers]
            if any(member) and not all(member):
                return
        a0, a1 = self.slab.range_of(ups)
        b0, b1 = self.slab.range_of(mid)
        if a1 != b0 or a1 <= a0:
            return
        self.exchange.early_range = (a0, b1)

        def early(module, gin, gout):
            if self.exchange._armed:
                self.slab.flush()
                self.exchange.launch_early()

        mid.register_full_backward_hook(early)

    # -- loss scale (fp16) ---------------------------------------------------------------------------
    GROWTH_INTERVAL = 2000  # torch.cuda.amp.GradScaler's default

    def _watch_overflow(self):
        """Ship the step's overflow flag to pinned host memory without waiting for it; LossScaler.begin_step reads it
        LAG steps later, behind a wait on the event recorded here (complete by then: the wait costs nothing)."""
        if self._flag_slots is None:
            return
        host, ev = self._flag_slots[self._flag_turn]
        self._flag_turn = (self._flag_turn + 1) % len(self._flag_slots)
        host.copy_(self.opt.norm[1:2], non_blocking=True)
        ev.record()

        def read(host=host, ev=ev):
            ev.synchronize()
            return float(host[0])

        self.scaler.watch(read)

    def _poll_overflow(self):
        """GradScaler's scale update at a fixed lag of LossScaler.LAG steps (identical on every rank).  A changed scale is
        picked up by this very step; a recorded hipGraph is re-recorded (the scale is baked into it)."""
        before = self.scaler.scale
        if self.scaler.begin_step() and self.scaler.scale < before and not self._warned_overflow:
            self._warned_overflow = True
            import warnings

            warnings.warn(f"LoraTrainer: non-finite fp16 gradients — the step was skipped and the loss scale "
                          f"lowered to {self.scaler.scale:g} (GradScaler semantics)")

    def _fingerprint(self):
        """Everything a recorded step has baked in besides the shapes: scalars passed as kernel arguments and the
        addresses / versions of the frozen operands the caches hand to the kernels."""
        layers = self.slab.layers
        # (the second linear layer of a hooked FeedForward hands cached W2 / W2ᵀ copies to the fused gate backward as well)
        ffs = self.__dict__.get("_ff_modules")
        if ffs is None:  # the module tree is walked once; the per-step check touches the 16 feed-forward blocks only
            ffs = self._ff_modules = [m for m in self.unet.modules()
                                      if m.__class__.__name__ == "FeedForward" and hasattr(m, "net") and len(m.net) == 3]
        ff2 = tuple((m.net[2].weight.data_ptr(), m.net[2].weight._version) for m in ffs if "forward" in m.__dict__)
        return (self.loss_scale, self.v_prediction, tuple(float(l.scale) for l in layers),
                tuple((l.linear.weight.data_ptr(), l.linear.weight._version) for l in layers), ff2)

    def _scheduled_lr_factor(self) -> float:
        """λ for the optimizer launch of this step, and the scheduler's own step() before or after it."""
        n = self.scheduler_steps_per_call
        if self.scheduler_steps_first:
            self.scheduler_epoch += n
            return float(self.lr_lambda(self.scheduler_epoch))
        self.scheduler_epoch += n
        return float(self.lr_lambda(self.scheduler_epoch - n))

    def get_last_lr(self) -> List[float]:
        """`lr_scheduler.get_last_lr()` (what the trainers log, train_lora_dreambooth.py:959): one value per param group."""
        lam = float(self.lr_lambda(self.scheduler_epoch))
        return [g["lr"] * lam for g in self.opt.groups]

    # -- one step ---------------------------------------------------------------------------------
    def step(self, latents, noise, timesteps, encoder_hidden_states=None, *, with_prior_preservation=False,
             prior_loss_weight=1.0, mask=None, seed: Optional[int] = None, input_ids=None, t_multiplier: float = 1.0):
        """latents fp32 [B,4,h,w] on the device.  Conditioning: `encoder_hidden_states` [B,L,D] — or `input_ids` [B,L]
        (int64), in which case the step itself runs `text_encoder(input_ids)[0]` as the reference's loop does
        (train_lora_dreambooth.py:840); with a LoRA text encoder that is what makes the step recordable.  Noise: either
        pass `noise` (fp32, like latents) and `timesteps` (int64 [B]) — the caller drew them, as the reference does — or
        pass None for both and a `seed`: the step then draws them on the device (Philox keyed by (seed, optimizer step),
        identical on every rank) inside the prologue kernel, timesteps uniform on [0, int(1000·t_multiplier)) — the PTI loop's
        `t_mutliplier` (cli_lora_pti.py:176,190-195).  `mask`: raw [B,1,8h,8w] mask of cli_lora_pti.py:222-247."""
        self._n_timesteps = max(1, int(self.sqrt_acp.numel() * float(t_multiplier)))
        if (encoder_hidden_states is None) == (input_ids is None):
            raise ValueError("pass exactly one of encoder_hidden_states and input_ids")
        if input_ids is not None and self.text_encoder is None:
            raise ValueError("input_ids given but the trainer has no text encoder")
        if self.token_table is not None and input_ids is not None and input_ids.device.type == "cpu":
            self.token_table.check_ids(input_ids)  # (free on the host; a replayed step cannot check device-resident ids)
        self._poll_overflow()
        recordable = not self.trains_text_encoder or input_ids is not None
        if self.capture_graph and recordable:
            return self._step_graph(latents, noise, timesteps, encoder_hidden_states, input_ids, mask,
                                    with_prior_preservation, prior_loss_weight, seed)
        return self._step_eager(latents, noise, timesteps, encoder_hidden_states, input_ids, with_prior_preservation,
                                prior_loss_weight, mask, seed)

    def _conditioning(self, encoder_hidden_states, input_ids):
        if input_ids is None:
            return encoder_hidden_states.to(self.dtype)
        if self.trains_text_encoder:
            return self.text_encoder(input_ids)[0].to(self.dtype)
        with torch.no_grad():  # frozen encoder: train_lora_dreambooth.py:608-609
            return self.text_encoder(input_ids)[0].to(self.dtype)

    def _forward_backward(self, noisy, target, timesteps, ehs, prior, prior_weight, raw_mask):
        """UNet forward → fused loss → backward → batched factor gradients + fold.  Shared by both launch modes."""
        pred = self.unet(noisy, timesteps, ehs).sample
        rows = pred.shape[0]
        n_inst, n_prior = (rows // 2, rows // 2) if prior else (rows, 0)
        m = None
        if raw_mask is not None:
            m = nat.lora_mask_prepare(raw_mask, pred.shape[2], pred.shape[3])
        pred_c = pred if pred.is_contiguous() else pred.contiguous()
        loss, dpred = nat.ddpm_mse_fwd_bwd(pred_c, target, m, n_inst, n_prior, prior_weight, self.loss_scale)
        pred_c.backward(dpred)
        self.slab.flush()  # factor gradients of every layer that ran: batched launch + ordered fold into the slab
        return loss

    def _raw_mask(self, mask, latents):
        if mask is None:
            return None
        rows, h, w = latents.shape[0], latents.shape[2], latents.shape[3]
        return mask.to(self.device).reshape(rows, 1, h * 8, w * 8).float().contiguous()

    def _step_eager(self, latents, noise, timesteps, encoder_hidden_states, input_ids, with_prior_preservation,
                    prior_loss_weight, mask, seed):
        """Host-launched step.  Exchange: two buckets with the early one overlapping backward where that is possible
        (SlabExchange / _install_bucket_hook) — unless a recorded step was asked for (exchange.single)."""
        self.slab.zero_grad()
        self.slab.repack()  # packed compute-dtype factors follow the fp32 masters (also after external edits)
        if noise is None:
            if seed is None:
                raise ValueError("pass noise and timesteps, or a seed for the on-device draw")
            noisy, target, timesteps = nat.ddpm_noise_prologue(latents, self.sqrt_acp, self.sqrt_1macp, self.dtype, seed,
                                                               self.opt.step_count, self.v_prediction, self._n_timesteps)
        else:
            noisy, target = nat.ddpm_add_noise(latents, noise, timesteps, self.sqrt_acp, self.sqrt_1macp, self.dtype,
                                               self.v_prediction)
        self.exchange.arm()
        if self.token_table is not None:
            self.token_table.begin_pass()
        ehs = self._conditioning(encoder_hidden_states, input_ids)
        loss = self._forward_backward(noisy, target, timesteps, ehs, with_prior_preservation, prior_loss_weight,
                                      self._raw_mask(mask, latents))
        tail = self.tail_events
        if tail is not None:  # (bench.py: what the step still has to do once backward has finished — with an early bucket its
            tail.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))  # all-reduce is behind us)
            tail[-1][0].record()
        self.exchange.finish()
        if self.token_table is not None:
            self.token_table.collect(self.pg, self.world if self.exchange.active else 1)
        self.opt.step(grad_mul=1.0 / (self.world * self.loss_scale), lr_mul=self._scheduled_lr_factor())
        self._watch_overflow()
        self.slab.repack()  # forwards outside step() (sampling, evaluation, saving merged weights) see the new factors
        if tail is not None:
            tail[-1][1].record()
        return loss

    # -- the same step with forward+backward replayed from a hipGraph -----------------------------------
    def _graph_body(self, st):
        """What is captured: reads only static buffers, leaves the gradient slab and the loss behind."""
        if st["draw"]:
            noisy, target = st["noisy"], st["target"]
        else:
            noisy, target = nat.ddpm_add_noise(st["latents"], st["noise"], st["timesteps"], self.sqrt_acp,
                                               self.sqrt_1macp, self.dtype, self.v_prediction)
        if self.token_table is not None:
            self.token_table.begin_pass()  # (runs while recording: the rows buffer it ends up holding is the recording's own)
        ehs = self._conditioning(st["ehs"], st["ids"])
        st["loss"] = self._forward_backward(noisy, target, st["timesteps"], ehs, st["prior"], st["prior_weight"], st["mask"])

    def _graph_inputs(self, st, latents, noise, timesteps, ehs, ids, mask, seed):
        if st["draw"]:
            if seed is None:
                raise ValueError("pass noise and timesteps, or a seed for the on-device draw")
            noisy, target, t = nat.ddpm_noise_prologue(latents, self.sqrt_acp, self.sqrt_1macp, self.dtype, seed,
                                                       self.opt.step_count, self.v_prediction, self._n_timesteps)
            st["noisy"].copy_(noisy)
            st["target"].copy_(target)
            st["timesteps"].copy_(t)
        else:
            st["latents"].copy_(latents)
            st["noise"].copy_(noise)
            st["timesteps"].copy_(timesteps)
        if ids is None:
            st["ehs"].copy_(ehs)  # casts to the compute dtype
        else:
            st["ids"].copy_(ids)
        if mask is not None:
            st["mask"].copy_(self._raw_mask(mask, latents))
        self.slab.zero_grad()
        self.slab.repack()

    def _step_graph(self, latents, noise, timesteps, ehs, ids, mask, prior, prior_weight, seed):
        cond_shape = tuple(ehs.shape) if ids is None else ("ids",) + tuple(ids.shape)
        key = (tuple(latents.shape), cond_shape, bool(prior), float(prior_weight), noise is None, mask is not None)
        fp = self._fingerprint()
        st = self._graph
        if st is None or st["key"] != key or st["fp"] != fp:
            self._graph = st = None  # drop the old recording (and the operand buffers it pins) before making a new one
            rows, h, w = latents.shape[0], latents.shape[2], latents.shape[3]
            st = {"key": key, "fp": fp, "draw": noise is None, "prior": bool(prior), "prior_weight": float(prior_weight),
                  "latents": torch.empty_like(latents, dtype=torch.float32),
                  "noise": torch.empty_like(latents, dtype=torch.float32),
                  "timesteps": torch.empty(latents.shape[0], dtype=torch.int64, device=self.device),
                  "noisy": torch.empty_like(latents, dtype=self.dtype), "target": torch.empty_like(latents, dtype=self.dtype),
                  "ehs": None if ids is not None else torch.empty_like(ehs, dtype=self.dtype),
                  "ids": None if ids is None else torch.empty_like(ids, device=self.device),
                  "mask": None if mask is None else torch.empty((rows, 1, h * 8, w * 8), dtype=torch.float32,
                                                                device=self.device),
                  "graph": None}
            self._graph_inputs(st, latents, noise, timesteps, ehs, ids, mask, seed)
            try:
                # warm up on a side stream (solver searches, lazy initialisation, allocator), then record
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._graph_body(st)
                torch.cuda.current_stream().wait_stream(side)
                self.slab.prepare_recording()  # pinned plan buffers the recording will own (never allocated inside it)
                g = torch.cuda.CUDAGraph()
                # With a process group alive, its watchdog thread polls HIP events every now and then; under the default
                # "global" capture mode such a call from ANOTHER thread invalidates the recording (a race that shows up
                # in a few percent of the captures).  "thread_local" keeps the check for this thread only.
                mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
                with torch.cuda.graph(g, capture_error_mode=mode):
                    self._graph_body(st)
                st["graph"] = g
                st["grad_plans"] = self.slab.take_recording_plans()  # host + device plan of the one-launch factor gradients
                # the (ids, gradient rows) buffers the RECORDING writes on every replay belong to the recording, not to the
                # table: an eager step in between resets the table's list (begin_pass), a replay cannot refill it
                st["token_pending"] = list(self.token_table._pending) if self.token_table is not None else None
            except Exception as exc:  # keep training: this trainer falls back to host-launched steps for good
                import warnings

                warnings.warn(f"LoraTrainer: hipGraph capture failed ({exc!r}); continuing with host-launched steps")
                self.capture_graph, self._graph = False, None
                # exchange.single stays set: this rank keeps issuing the one whole-slab all-reduce its peers' replays issue
                return self._step_eager(latents, noise, timesteps, ehs, ids, prior, prior_weight, mask, seed)
            self._graph = st
            # the warm-up passes each folded this step's gradients into the slab: clear it, then replay once so that
            # every step (including the first) is produced by the same recorded kernels
            self.slab.zero_grad()
        else:
            self._graph_inputs(st, latents, noise, timesteps, ehs, ids, mask, seed)
        st["graph"].replay()
        tail = self.tail_events
        if tail is not None:  # (bench.py: device time of the host-launched tail — exchange, clip + AdamW, re-pack)
            tail.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            tail[-1][0].record()
        self.exchange.finish()
        if self.token_table is not None:
            self.token_table._pending = list(st["token_pending"])
            self.token_table.collect(self.pg, self.world if self.exchange.active else 1)
        self.opt.step(grad_mul=1.0 / (self.world * self.loss_scale), lr_mul=self._scheduled_lr_factor())
        self._watch_overflow()
        self.slab.repack()
        if tail is not None:
            tail[-1][1].record()
        return st["loss"].clone()


def flat_lora_state(model: nn.Module, targets=None) -> torch.Tensor:
    """[up0, down0, up1, ...] flattened — the `.pt` file order."""
    return torch.cat([t.detach().float().reshape(-1) for l in lora_layers(model)
                      for t in (l.lora_up.weight, l.lora_down.weight)])
