"""Autograd front of the HIP hot path.

`lora_linear(module, x)` is what `LoraInjectedLinear.forward` calls: one fused forward kernel, and in the
backward one fused dX kernel plus one factor-gradient kernel (reference arithmetic: lora_diffusion/lora.py:49-50
and its autograd).  `ddpm_mse_loss` is the fused loss of training_scripts/train_lora_dreambooth.py:855-875 and
lora_diffusion/cli_lora_pti.py:222-247.

Frozen operands are cached per module in the layouts the kernels want — W in the compute dtype and its
transpose Wᵀ (so the backward contraction is contiguous too).  That costs 2× the frozen-weight bytes, which is
nothing on a 288 GB part, and removes the per-call fp32→fp16 weight cast the reference pays under autocast.
The cache is keyed on the weight's storage pointer, version counter, dtype and device, so `.to()`, `.half()`,
`weight_apply_lora` or assigning a new Parameter invalidate it.
"""
import warnings

import torch
from torch.autograd.function import once_differentiable

from . import _native as nat

_warned_trainable_base = False


def _compute_dtype(weight: torch.Tensor) -> torch.dtype:
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return weight.dtype


def _frozen_operands(module, cdtype: torch.dtype, need_wt: bool):
    """Returns (W, Wᵀ|None, bias|None) in `cdtype`, building them once per (weight state, dtype)."""
    lin = module.linear
    w = lin.weight
    b = lin.bias
    key = (w.data_ptr(), w._version, w.dtype, w.device, cdtype,
           None if b is None else (b.data_ptr(), b._version))
    cache = module.__dict__.get("_dfa_cache")
    if cache is None or cache["key"] != key:
        wd = w.detach()
        if not wd.is_contiguous():
            wd = wd.contiguous()
        cache = {
            "key": key,
            "w": wd if wd.dtype == cdtype else nat.lora_cast_matrix(wd, cdtype, False),
            "wt": None,
            "bias": None if b is None else b.detach().to(cdtype).contiguous(),
        }
        module.__dict__["_dfa_cache"] = cache
    if need_wt and cache["wt"] is None:
        cache["wt"] = nat.lora_cast_matrix(cache["w"], cdtype, True)
    return cache["w"], cache["wt"], cache["bias"]


def invalidate_weight_cache(model: torch.nn.Module) -> None:
    """Drops every cached W/Wᵀ copy under `model` (call after mutating frozen weights through `.data`)."""
    for m in model.modules():
        m.__dict__.pop("_dfa_cache", None)


def _as_f32(p: torch.Tensor) -> torch.Tensor:
    p = p.detach()
    if p.dtype != torch.float32:
        p = p.float()
    return p if p.is_contiguous() else p.contiguous()


class _LoraLinearFn(torch.autograd.Function):
    """y = x·Wᵀ + b + s·(x·Aᵀ)·Bᵀ with grads for x, A (down) and B (up) only."""

    @staticmethod
    def forward(ctx, x, down, up, w, wt, bias, scale, grad_sink, packed):
        K = w.shape[1]
        N = w.shape[0]
        x2 = x.reshape(-1, K)
        if x2.dtype != w.dtype:
            x2 = x2.to(w.dtype)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        a = _as_f32(down)
        b = _as_f32(up)
        # packed factors: the trainer keeps them current for the whole slab (one launch per optimizer step);
        # otherwise they are cast here, once per call, like autocast casts lora_down/lora_up in the reference
        packs = packed if packed is not None else nat.lora_pack_factors(a, b, w.dtype)
        y2, t = nat.lora_linear_fwd(x2, w, bias, a, b, scale, packs)
        ctx.save_for_backward(x2, a, b, t)
        ctx.packs = packs
        ctx.wt = wt
        ctx.scale = float(scale)
        ctx.x_shape = x.shape
        ctx.x_dtype = x.dtype
        ctx.factor_dtypes = (down.dtype, up.dtype)
        ctx.grad_sink = grad_sink
        ctx.auto_sink = _auto_sink_for(down, up) if grad_sink is None else None
        ctx.params = (down, up) if ctx.auto_sink is not None else None
        return y2.view(*x.shape[:-1], N)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, a, b, t = ctx.saved_tensors
        N = b.shape[0]
        dy2 = dy.reshape(-1, N)
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        return (*_lora_backward(ctx, x2, a, b, t, dy2), None, None, None, None, None, None)


class _AutoSink:
    """Drop-in mode (no trainer.LoraSlab): where the factor gradients of a plain `loss.backward()` go.

    An unchanged reference trainer (train_lora_dreambooth.py:877) never builds a slab, so every wrapped layer used to launch
    its own two reductions (+ workspace, fold and two accumulations: ~7 small launches × 144 layers, latency-bound).  Here the
    backward of a layer only DEFERS its two problems; when the autograd engine finishes the pass (`queue_callback`, the hook
    DDP's reducer uses) ONE `lora_grad_batched` call launches all of them, one fold sums the row-block partials, and the
    results are handed to the Parameters' `.grad` (set, or accumulated into an existing one — gradient accumulation keeps
    working).  Not used when a process group is alive (DDP all-reduces what AccumulateGrad hands it: those steps keep the
    per-layer launches that return real gradient tensors), when a Parameter carries hooks, or with DFA_DEFER_GRADS=0.
    `torch.autograd.grad(loss, lora_params)` is not a `.backward()`: it sees None for deferred inputs and raises unless
    allow_unused is set — use DFA_DEFER_GRADS=0 for such callers."""

    def __init__(self, device):
        self.device = device
        self.items = []
        self.armed = False
        self.partials = None
        self.tables = {}

    def defer(self, dy2, x2, t, u, scale, down, up, dtypes):
        """One layer: gB = s·dYᵀ·T → up, gA = s·Uᵀ·X → down."""
        M, N = dy2.shape
        K, r = x2.shape[1], t.shape[1]
        self.defer_problem(dy2, 0, N, N, t, 0, r, r, False, M, scale, [(up, dtypes[1])])
        self.defer_problem(x2, 0, K, K, u, 0, r, r, True, M, scale, [(down, dtypes[0])])

    def defer_problem(self, S, s_off, s_stride, C, P, p_off, p_stride, rg, out_kn, M, scale, targets):
        """G[c, j] = scale·Σ_m S[m, s_off + c]·P[m, p_off + j] for len(targets)·rg rank columns; rank group i (rg columns) is
        the whole gradient of the Parameter targets[i][0] ([C, rg] for an `up`, [rg, C] for a `down` — out_kn), handed over
        in targets[i][1].  Operands may be column slices of wider buffers (grouped projections)."""
        self.items.append((S, s_off, s_stride, C, P, p_off, p_stride, rg, out_kn, M, scale, targets))
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        items, self.items, self.armed = self.items, [], False
        if not items:
            return
        offs, total = [], 0
        for it in items:
            C, rg, targets = it[3], it[7], it[11]
            offs.append(total)
            total += len(targets) * rg * C
        stride = (total + 3) // 4 * 4
        if self.partials is None or self.partials.shape[1] < stride:
            self.partials = torch.empty((nat.GRAD_MAX_BLOCKS, stride), dtype=torch.float32, device=self.device)
        part, pstride = self.partials, self.partials.shape[1]
        grads = torch.empty(stride, dtype=torch.float32, device=self.device)
        base = part.data_ptr()
        by_dtype, rows = {}, []
        for off, (S, s_off, s_stride, C, P, p_off, p_stride, rg, out_kn, M, scale, targets) in zip(offs, items):
            n = len(targets)
            outs = [base + 4 * (off + i * rg * C) for i in range(n)]
            by_dtype.setdefault(S.dtype, []).append(
                nat.grad_problem(S, s_off, s_stride, C, P, p_off, p_stride, n * rg, outs, rg, out_kn, pstride, M, scale))
            rows.append([off, n * rg * C, nat.grad_row_blocks(M), 0])
        for dt, probs in by_dtype.items():
            nat.lora_grad_batched(probs, dt, self.device)
        key = tuple(map(tuple, rows))
        table = self.tables.get(key)  # (one host→device copy per distinct set of layer shapes, not one per step)
        if table is None:
            table = self.tables[key] = torch.tensor(rows, dtype=torch.int64).to(self.device)
        nat.lora_fold_partials(table, len(rows), max(r_[1] for r_ in rows), part, pstride, grads, False)
        for off, it in zip(offs, items):
            C, rg, out_kn, targets = it[3], it[7], it[8], it[11]
            for i, (p, dt) in enumerate(targets):
                if not p.requires_grad:
                    continue  # (a factor frozen by the caller: autograd would not have produced its gradient either)
                g = grads[off + i * rg * C: off + (i + 1) * rg * C].view((rg, C) if out_kn else (C, rg))
                if dt != torch.float32:
                    g = g.to(dt)
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad += g


class PackRegistry:
    """Drop-in mode (no trainer.LoraSlab): the packed compute-dtype factors of ALL layers one `inject_trainable_lora` call
    wrapped, refreshed by ONE `lora_pack_items` launch when a forward finds its own factors changed (an optimizer step, a
    loaded file, `.to()`) instead of one `lora_pack_factors` launch + two allocations per layer call.  What the reference pays
    at this point is autocast's per-call cast of `lora_down/lora_up.weight` (lora.py:50 under train_lora_dreambooth.py:489-494).
    It also owns the packed operands of the GROUPED projections an unchanged trainer gets (groups.QKVGroup / CtxKVGroup built
    by the attention switch, attention.set_use_hip_attention): their rows ride in the same launch."""

    def __init__(self, modules):
        self.modules = [m for m in modules if m.lora_down.weight.shape[0] <= 16]
        self.index = {id(m): i for i, m in enumerate(self.modules)}
        self.state = None  # (dtype, device, per-module (ptr, version) pairs) the packed buffer was built from
        self.views = None
        self.table = None
        self.ptrs = None
        self.groups = []

    def add_group(self, grp) -> bool:
        """A grouped projection whose members all live in this registry; its operands are laid out at the next refresh."""
        if not all(id(l) in self.index for l in grp.layers):
            return False
        grp.registry = self
        self.groups.append(grp)
        self.table = None
        return True

    def drop_groups(self):
        for grp in self.groups:
            grp.registry = None
        self.groups = []
        self.table = None

    @staticmethod
    def _sig(m):
        d, u = m.lora_down.weight, m.lora_up.weight
        return (d.data_ptr(), d._version, u.data_ptr(), u._version)

    def _current(self, members, cdtype, device) -> bool:
        st = self.state
        return (st is not None and self.table is not None and st[0] == cdtype and st[1] == device and
                all(st[2][self.index[id(m)]] == self._sig(m) for m in members))

    def get(self, module, cdtype):
        i = self.index.get(id(module))
        if i is None:
            return None
        d, u = module.lora_down.weight, module.lora_up.weight
        if d.dtype != torch.float32 or u.dtype != torch.float32 or not d.is_cuda:
            return None  # (factors held in another dtype are cast per call, as before)
        if not self._current((module,), cdtype, d.device) and not self._repack(cdtype, d.device):
            return None
        return self.views[i]

    def ensure(self, members, cdtype) -> bool:
        """True when the packed operands (a group's included) are current for `members` — refreshing them if needed."""
        d = members[0].lora_down.weight
        if d.dtype != torch.float32 or not d.is_cuda:
            return False
        return self._current(members, cdtype, d.device) or self._repack(cdtype, d.device)

    def _repack(self, cdtype, device):
        from .groups import QKVGroup, bind_ctx_views, bind_qkv_views, ctx_pack_rows, qkv_pack_rows

        mods = self.modules
        if any(m.lora_down.weight.dtype != torch.float32 or m.lora_down.weight.device != device or
               not m.lora_down.weight.is_contiguous() or not m.lora_up.weight.is_contiguous() for m in mods):
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        ptrs = tuple((m.lora_down.weight.data_ptr(), m.lora_up.weight.data_ptr()) for m in mods)
        if self.table is None or self.ptrs != ptrs or self.views is None or self.views[0][0].dtype != cdtype:
            # element offsets relative to ONE base pointer: the factors live in separate allocations, the pack kernel adds a
            # signed 64-bit offset to its `params` argument
            base = mods[0].lora_down.weight

            def rel(p):
                return (p.data_ptr() - base.data_ptr()) // 4

            rows, views_at, off = [], [], 0
            for m in mods:
                r, K = m.lora_down.weight.shape
                N = m.lora_up.weight.shape[0]
                rows.append([rel(m.lora_down.weight), 0, K, r, off, K, off + 16 * K, 16])
                rows.append([rel(m.lora_up.weight), 1, N, r, off + 32 * K, N, off + 32 * K + 16 * N, 16])
                views_at.append((off, K, N))
                off += 32 * (K + N)
            binds = []
            for grp in self.groups:
                src = [(rel(l.lora_up.weight), rel(l.lora_down.weight)) for l in grp.layers]
                qkv = isinstance(grp, QKVGroup)
                g_rows, spec, used = (qkv_pack_rows if qkv else ctx_pack_rows)(grp, src, off)
                rows += g_rows
                binds.append((grp, spec, bind_qkv_views if qkv else bind_ctx_views, qkv))
                off += used
            self.packed = torch.zeros(off, dtype=cdtype, device=device)  # zeroed: block-diagonal groups fill their own slots only
            self.table = torch.tensor(rows, dtype=torch.int64).to(device)
            self.maxlen = max(r_[2] for r_ in rows)
            self.views = [(self.packed[o:o + 32 * K], self.packed[o + 32 * K:o + 32 * (K + N)]) for o, K, N in views_at]
            for grp, spec, bind, qkv in binds:
                bind(grp, self.packed, spec)
                if qkv:
                    grp.Fb_part = [self.views[self.index[id(l)]][1][:16 * grp.N] for l in grp.layers]
            self.ptrs = ptrs
            self.base = base
        nat.lora_pack_items(self.table, self.table.shape[0], self.maxlen, self.base.detach(), self.packed)
        self.state = (cdtype, device, tuple(self._sig(m) for m in mods))
        return True


def register_pack_group(modules) -> None:
    """Called by `inject_trainable_lora` with the modules it wrapped (core.py)."""
    import os

    if os.environ.get("DFA_PACK_REGISTRY", "1") == "0":
        return
    modules = list(modules)
    if len(modules) < 2:
        return
    reg = PackRegistry(modules)
    for m in reg.modules:
        m.__dict__["_dfa_packreg"] = reg


_auto_sinks = {}


def _auto_sink_for(down, up):
    """The drop-in sink of the tensors' device, or None when deferring is not safe for these Parameters (see _AutoSink)."""
    import os

    import torch.distributed as dist

    if os.environ.get("DFA_DEFER_GRADS", "1") == "0" or (dist.is_available() and dist.is_initialized()):
        return None
    for p in (down, up):
        if not isinstance(p, torch.nn.Parameter) or not p.is_leaf or p._backward_hooks or \
                getattr(p, "_post_accumulate_grad_hooks", None):
            return None
    sink = _auto_sinks.get(down.device)
    if sink is None:
        sink = _auto_sinks[down.device] = _AutoSink(down.device)
    return sink


def _lora_backward(ctx, x2, a, b, t, dy2):
    """dX (one fused kernel) and the two factor gradients (deferred to the slab's batched launch in trainer mode) for
    dy2 [M,N]; shared by the plain and the GEGLU-gated autograd fronts."""
    need_dx = ctx.needs_input_grad[0]
    need_factors = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
    if need_dx and ctx.wt is None:
        raise RuntimeError("lora_linear backward: Wᵀ operand was not prepared in forward")
    dx2, u = nat.lora_linear_bwd_input(dy2, ctx.wt if need_dx else None, a, b, ctx.scale, need_dx, ctx.packs)
    g_down = g_up = None
    if need_factors:
        sink = ctx.grad_sink
        if sink is not None:
            # trainer mode: nothing is launched here — the two reductions join the slab's batched gradient launch
            # after backward, their row-block partials land in the model-wide partial slab (= the RCCL buffer's twin)
            sink.defer_layer(dy2, x2, t, u, ctx.scale, need_dx)
        elif ctx.auto_sink is not None:
            # drop-in mode: the two reductions join ONE batched launch at the end of this backward pass (_AutoSink)
            down, up = ctx.params
            ctx.auto_sink.defer(dy2, x2, t, u, ctx.scale, down, up, ctx.factor_dtypes)
        else:
            g_down = torch.zeros_like(a)
            g_up = torch.zeros_like(b)
            nat.lora_linear_bwd_params(dy2, x2, t, u, g_down, g_up, ctx.scale)
            if ctx.factor_dtypes[0] != torch.float32:
                g_down = g_down.to(ctx.factor_dtypes[0])
            if ctx.factor_dtypes[1] != torch.float32:
                g_up = g_up.to(ctx.factor_dtypes[1])
    dx = None
    if need_dx:
        dx = dx2.view(ctx.x_shape)
        if dx.dtype != ctx.x_dtype:
            dx = dx.to(ctx.x_dtype)
    return dx, g_down, g_up


class _LoraGegluFn(torch.autograd.Function):
    """out = h·gelu(g), [h | g] = x·Wᵀ + b + s·(x·Aᵀ)·Bᵀ — the `proj` LoraInjectedLinear of a GEGLU module together with the
    gate of diffusers GEGLU.forward, forward in ONE launch (the gate sits in the GEMM epilogue).  y = [h | g] is written only
    when a backward pass will need it; backward = gate backward (one streaming kernel) + the LoRA backward."""

    @staticmethod
    def forward(ctx, x, down, up, w, wt, bias, scale, grad_sink, packed):
        K = w.shape[1]
        x2 = x.reshape(-1, K)
        if x2.dtype != w.dtype:
            x2 = x2.to(w.dtype)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        a = _as_f32(down)
        b = _as_f32(up)
        packs = packed if packed is not None else nat.lora_pack_factors(a, b, w.dtype)
        need = any(ctx.needs_input_grad[:3])
        res = nat.lora_linear_geglu_fwd(x2, w, bias, a.shape[0], scale, packs, need)
        if res is None:  # no fused kernel for this shape / dtype: the two launches it stands for
            y2, t = nat.lora_linear_fwd(x2, w, bias, a, b, scale, packs)
            out = nat.geglu_gate_fwd(y2)
        else:
            out, y2, t = res
        if need:
            ctx.save_for_backward(x2, a, b, t, y2)
        ctx.packs = packs
        ctx.wt = wt
        ctx.scale = float(scale)
        ctx.x_shape = x.shape
        ctx.x_dtype = x.dtype
        ctx.factor_dtypes = (down.dtype, up.dtype)
        ctx.grad_sink = grad_sink
        ctx.auto_sink = _auto_sink_for(down, up) if grad_sink is None else None
        ctx.params = (down, up) if ctx.auto_sink is not None else None
        return out.view(*x.shape[:-1], w.shape[0] // 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x2, a, b, t, y2 = ctx.saved_tensors
        d2 = dout.reshape(-1, dout.shape[-1])
        if d2.dtype != y2.dtype:
            d2 = d2.to(y2.dtype)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        dy2 = nat.geglu_gate_bwd(y2, d2)
        return (*_lora_backward(ctx, x2, a, b, t, dy2), None, None, None, None, None, None)


class _LoraProjGatedFn(torch.autograd.Function):
    """The `proj` LoraInjectedLinear of a GEGLU block for a caller that also owns what follows the gate (feed_forward below):
    ONE launch produces y = [h | g] — the differentiable output, whose gradient is the usual LoRA backward — and the gated
    activation h·gelu(g) as a non-differentiable by-product."""

    @staticmethod
    def forward(ctx, x, down, up, w, wt, bias, scale, grad_sink, packed):
        K = w.shape[1]
        x2 = x.reshape(-1, K)
        if x2.dtype != w.dtype:
            x2 = x2.to(w.dtype)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        a = _as_f32(down)
        b = _as_f32(up)
        packs = packed if packed is not None else nat.lora_pack_factors(a, b, w.dtype)
        res = nat.lora_linear_geglu_fwd(x2, w, bias, a.shape[0], scale, packs, True)
        if res is None:
            y2, t = nat.lora_linear_fwd(x2, w, bias, a, b, scale, packs)
            out = nat.geglu_gate_fwd(y2)
        else:
            out, y2, t = res
        ctx.save_for_backward(x2, a, b, t)
        ctx.packs = packs
        ctx.wt = wt
        ctx.scale = float(scale)
        ctx.x_shape = x.shape
        ctx.x_dtype = x.dtype
        ctx.factor_dtypes = (down.dtype, up.dtype)
        ctx.grad_sink = grad_sink
        ctx.auto_sink = _auto_sink_for(down, up) if grad_sink is None else None
        ctx.params = (down, up) if ctx.auto_sink is not None else None
        N = w.shape[0]
        out = out.view(*x.shape[:-1], N // 2)
        ctx.mark_non_differentiable(out)
        return y2.view(*x.shape[:-1], N), out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dout):
        x2, a, b, t = ctx.saved_tensors
        dy2 = dy.reshape(-1, b.shape[0])
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        return (*_lora_backward(ctx, x2, a, b, t, dy2), None, None, None, None, None, None)


class _GatedLinearFn(torch.autograd.Function):
    """z = gated @ W2ᵀ + b2 with gated = h·gelu(g) of y = [h | g] (diffusers FeedForward: net[2](net[0](x))), W2 / b2 frozen.
    The graph edge goes to y, not to `gated`: backward is ONE launch, dY = gate-backward(dz·W2, y) (`geglu_linear_bwd`) —
    the gate's backward sits in the epilogue of the linear layer's backward-input GEMM."""

    @staticmethod
    def forward(ctx, y, gated, w2, w2t, b2):
        ctx.save_for_backward(y, w2, w2t)
        return torch.nn.functional.linear(gated, w2, b2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        y, w2, w2t = ctx.saved_tensors
        y2 = y.reshape(-1, y.shape[-1])
        dz2 = dz.reshape(-1, dz.shape[-1])
        if dz2.dtype != y2.dtype:
            dz2 = dz2.to(y2.dtype)
        if not dz2.is_contiguous():
            dz2 = dz2.contiguous()
        dy = nat.geglu_linear_bwd(dz2, w2t, y2)
        if dy is None:  # no fused kernel for this shape / dtype: the two steps it stands for
            dy = nat.geglu_gate_bwd(y2, dz2 @ w2)
        return dy.view(y.shape), None, None, None, None


def _frozen_linear(lin, cdtype: torch.dtype):
    """(W, Wᵀ, bias) of a frozen nn.Linear in `cdtype`, cached on the module like _frozen_operands."""
    w, b = lin.weight, lin.bias
    key = (w.data_ptr(), w._version, w.dtype, w.device, cdtype, None if b is None else (b.data_ptr(), b._version))
    cache = lin.__dict__.get("_dfa_cache")
    if cache is None or cache["key"] != key:
        wd = w.detach()
        if not wd.is_contiguous():
            wd = wd.contiguous()
        wc = wd if wd.dtype == cdtype else nat.lora_cast_matrix(wd, cdtype, False)
        cache = {"key": key, "w": wc, "wt": nat.lora_cast_matrix(wc, cdtype, True),
                 "bias": None if b is None else b.detach().to(cdtype).contiguous()}
        lin.__dict__["_dfa_cache"] = cache
    return cache["w"], cache["wt"], cache["bias"]


def feed_forward_geglu(proj_module, lin2, x: torch.Tensor) -> torch.Tensor:
    """diffusers FeedForward with a GEGLU activation — `net[2](net[0](x))`, net[0] = GEGLU(proj), net[2] = frozen Linear — with
    BOTH halves of the gate inside GEMM epilogues: forward in the `proj` launch, backward in the launch that computes net[2]'s
    input gradient.  `proj_module` is the LoraInjectedLinear, `lin2` the nn.Linear."""
    y, gated = lora_linear(proj_module, x, gate="pair")
    w2, w2t, b2 = _frozen_linear(lin2, y.dtype)
    return _GatedLinearFn.apply(y, gated, w2, w2t, b2)


def lora_linear(module, x: torch.Tensor, gate: bool = False) -> torch.Tensor:
    """Fused LoraInjectedLinear forward (lora_diffusion/lora.py:49-50) on the HIP device; `gate`: see lora_linear_geglu."""
    global _warned_trainable_base
    lin, down, up = module.linear, module.lora_down.weight, module.lora_up.weight
    if not x.is_cuda or not lin.weight.is_cuda:
        raise RuntimeError(
            "LoraInjectedLinear.forward: the fused LoRA path runs only on a HIP device (MI355X); got input on "
            f"{x.device} and weight on {lin.weight.device}. Move the model and inputs to 'cuda' — there is no CPU fallback."
        )
    if lin.weight.requires_grad and torch.is_grad_enabled() and not _warned_trainable_base:
        _warned_trainable_base = True
        warnings.warn(
            "LoraInjectedLinear: the base weight has requires_grad=True, but this path treats W and b as frozen "
            "(no ∇W/∇b are produced). Call model.requires_grad_(False) before inject_trainable_lora as the "
            "reference trainers do."
        )
    cdtype = _compute_dtype(lin.weight)
    shared = module.__dict__.get("_dfa_shared")  # (group, member): projections of one input called one by one (CLIP q/k/v)
    if shared is not None and not gate and shared[0].usable(x, cdtype) and x.dim() >= 2:
        from .groups import shared_projection

        return shared_projection(shared[0], shared[1], x, cdtype)
    need_wt = torch.is_grad_enabled() and x.requires_grad
    w, wt, bias = _frozen_operands(module, cdtype, need_wt)
    sink = module.__dict__.get("_dfa_grad_sink")
    packed = module.__dict__.get("_dfa_packed")
    if packed is not None and packed[0].dtype != cdtype:
        packed = None
    if packed is None and sink is None:
        reg = module.__dict__.get("_dfa_packreg")
        if reg is not None:
            packed = reg.get(module, cdtype)  # drop-in mode: all layers' packed factors from one launch per update
    fn = _LoraProjGatedFn if gate == "pair" else (_LoraGegluFn if gate else _LoraLinearFn)
    return fn.apply(x, down, up, w, wt, bias, float(module.scale), sink, packed)


def lora_linear_geglu(module, x: torch.Tensor) -> torch.Tensor:
    """`hidden * gelu(gate)` of `module(x).chunk(2, -1)` for the `proj` LoraInjectedLinear of a GEGLU block (diffusers
    GEGLU.forward; LoRA target class "GEGLU", lora_diffusion/lora.py:53) with the gate inside the forward kernel's epilogue."""
    return lora_linear(module, x, gate=True)


class _DDPMLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, n_inst, n_prior, prior_weight):
        p = pred if pred.is_contiguous() else pred.contiguous()
        t = target.detach()
        if t.dtype != p.dtype:
            t = t.to(p.dtype)
        if not t.is_contiguous():
            t = t.contiguous()
        loss, dpred = nat.ddpm_mse_fwd_bwd(p, t, mask, n_inst, n_prior, prior_weight, 1.0,
                                           want_grad=ctx.needs_input_grad[0])
        if dpred is not None:
            ctx.save_for_backward(dpred)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.to(dpred.dtype), None, None, None, None, None


def ddpm_mse_loss(pred, target, *, with_prior_preservation=False, prior_loss_weight=1.0, mask=None):
    """DDPM noise-prediction loss in one fused pass.

    Plain:  mean((pred.float()-target.float())²)                       (train_lora_dreambooth.py:875)
    Prior:  batch halves → instance mean-of-means + w·prior mean        (train_lora_dreambooth.py:855-873)
    Mask:   `mask` is the RAW [B,1,8h,8w] mask of cli_lora_pti.py:222-241; it is resized (nearest), +0.05,
            mean-normalised on the device and applied to pred and target (cli_lora_pti.py:243-247).
    Returns a 0-d fp32 tensor that backpropagates into `pred`.
    """
    if not pred.is_cuda:
        raise RuntimeError("ddpm_mse_loss runs only on a HIP device; there is no CPU fallback")
    rows = pred.shape[0]
    if with_prior_preservation:
        if rows % 2 != 0:
            raise ValueError("prior preservation needs an even batch (instance rows then class rows)")
        n_inst = n_prior = rows // 2
    else:
        n_inst, n_prior = rows, 0
    m = None
    if mask is not None:
        raw = mask.to(pred.device).reshape(rows, 1, pred.shape[2] * 8, pred.shape[3] * 8).float().contiguous()
        m = nat.lora_mask_prepare(raw, pred.shape[2], pred.shape[3])
    return _DDPMLossFn.apply(pred, target, m, n_inst, n_prior, float(prior_loss_weight))
