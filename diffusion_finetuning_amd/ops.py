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
import os
import warnings

import torch
import torch.distributed as dist
from torch.autograd.function import once_differentiable

from . import _native as nat
from ._fastattr import factor_weights, frozen_linear, linear_params

_warned_trainable_base = False


def _compute_dtype(weight: torch.Tensor) -> torch.dtype:
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return weight.dtype


def _frozen_operands(module, cdtype: torch.dtype, need_wt: bool, w=None, b=None):
    """Returns (W, Wᵀ|None, bias|None) in `cdtype`, building them once per (weight state, dtype); `w` / `b`: the frozen
    Parameters when the caller has looked them up already."""
    if w is None:
        w, b = frozen_linear(module)
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
        y, saved = _lora_forward(ctx, x, down, up, w, wt, bias, scale, grad_sink, packed)
        ctx.save_for_backward(*saved)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return (*_lora_backward_of(ctx, dy, *ctx.saved_tensors), None, None, None, None, None, None)


def _lora_forward(ctx, x, down, up, w, wt, bias, scale, grad_sink, packed):
    """Forward body of one fused LoRA linear for the autograd node (or the LoraTail of a node) `ctx`: the launch, and on `ctx`
    everything but tensors that `_lora_backward` reads.  Returns (y, (x2, a, b, t)) — the tuple is what the node must save."""
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
    ctx.packs = packs
    ctx.wt = wt
    ctx.scale = float(scale)
    ctx.x_shape = x.shape
    ctx.x_dtype = x.dtype
    ctx.factor_dtypes = (down.dtype, up.dtype)
    ctx.grad_sink = grad_sink
    ctx.auto_sink = _auto_sink_for(down, up) if grad_sink is None else None
    ctx.params = (down, up) if ctx.auto_sink is not None else None
    return y2.view(*x.shape[:-1], N), (x2, a, b, t)


def _lora_backward_of(ctx, dy, x2, a, b, t):
    """Backward body: (dx, g_down, g_up) for the incoming gradient `dy` of the layer's output."""
    N = b.shape[0]
    dy2 = dy.reshape(-1, N)
    if dy2.dtype != x2.dtype:
        dy2 = dy2.to(x2.dtype)
    if not dy2.is_contiguous():
        dy2 = dy2.contiguous()
    return _lora_backward(ctx, x2, a, b, t, dy2)


class LoraTail:
    """One LoraInjectedLinear run INSIDE another autograd node, behind that node's own kernels — `to_out[0]` of an attention
    module behind the attention core (groups._QKVAttnFn, groups._CtxAttnKVFn).  Same launches as `_LoraLinearFn`, one
    `Function.apply` and one backward node fewer per layer (≈ 25 µs of host time on the unchanged-trainer route, which is
    host-bound).  The object carries what `lora_linear` looked up for the layer (`operands`) and, between forward and
    backward, the non-tensor state `_lora_backward` reads; the tensors to keep are returned to the node, which saves them
    with its own (`ctx.save_for_backward`: the version checks of the factors stay in force).  The two factor Parameters are
    inputs of the node (`factors`), so autograd tracks them exactly as it does for `_LoraLinearFn`."""

    def __init__(self, operands):
        self.operands = operands  # (down, up, w, wt, bias, scale, grad_sink, packed): what `lora_linear` hands `_LoraLinearFn`
        self.state = None

    @property
    def factors(self):
        return self.operands[0], self.operands[1]

    def forward(self, x, need_down: bool, need_up: bool):
        """(y, tensors the node must save) for the layer's input `x` (an internal tensor of the node)."""
        self.state = _TailState((True, need_down, need_up))
        return _lora_forward(self.state, x, *self.operands)

    def backward(self, dy, x2, a, b, t):
        """(dx, g_down | None, g_up | None): the gradient of the layer's input, and of its factors where they are not deferred."""
        return _lora_backward_of(self.state, dy, x2, a, b, t)


class _TailState:
    """The `ctx` a LoraTail shows `_lora_forward` / `_lora_backward`: plain attributes."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad


def lora_tail(module, cdtype: torch.dtype):
    """A LoraTail for `module` — or None when the layer is not a plain LoraInjectedLinear on the HIP device (a subclass, an
    overridden forward, forward hooks, a CLIP-style shared projection): such a layer is simply called."""
    from .core import LoraInjectedLinear

    if (type(module) is not LoraInjectedLinear or "forward" in module.__dict__ or module._forward_hooks or
            module._forward_pre_hooks or "_dfa_shared" in module.__dict__):
        return None
    w_param, b_param = frozen_linear(module)
    if not w_param.is_cuda or (w_param.requires_grad and torch.is_grad_enabled()):
        return None  # (lora_linear raises / warns for these: let it)
    return LoraTail(_layer_operands(module, cdtype, torch.is_grad_enabled(), w_param, b_param, *factor_weights(module)))


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
        self.items = []       # (S, P, _SinkProblem) of the pass being run
        self.armed = False
        self.partials = None
        self.shapes = {}      # everything of a problem but its two operand pointers → _SinkProblem
        self.plans = {}       # the _SinkProblem objects of a whole pass, in order → _SinkPlan

    def defer(self, dy2, x2, t, u, scale, down, up, dtypes):
        """One layer: gB = s·dYᵀ·T → up, gA = s·Uᵀ·X → down."""
        M, N = dy2.shape
        K, r = x2.shape[1], t.shape[1]
        self.defer_problem(dy2, 0, N, N, t, 0, r, r, False, M, scale, ((up, dtypes[1]),))
        self.defer_problem(x2, 0, K, K, u, 0, r, r, True, M, scale, ((down, dtypes[0]),))

    def defer_problem(self, S, s_off, s_stride, C, P, p_off, p_stride, rg, out_kn, M, scale, targets):
        """G[c, j] = scale·Σ_m S[m, s_off + c]·P[m, p_off + j] for len(targets)·rg rank columns; rank group i (rg columns) is
        the whole gradient of the Parameter targets[i][0] ([C, rg] for an `up`, [rg, C] for a `down` — out_kn), handed over
        in targets[i][1].  Operands may be column slices of wider buffers (grouped projections).
        A training loop defers the same problems every step, only on new activations: what does not change is kept as ONE
        object per distinct problem, so that a whole pass is recognised by the identity of its objects (flush)."""
        key = (s_off, s_stride, C, p_off, p_stride, rg, out_kn, M, scale, S.dtype, *[(id(p), dt) for p, dt in targets])
        prob = self.shapes.get(key)
        if prob is None:
            if len(self.shapes) >= 4096:  # (a caller that keeps changing shapes or Parameters: start over)
                self.shapes.clear()
                self.plans.clear()
            # (the object holds its Parameters: their ids in `key` cannot be reused while it lives)
            prob = self.shapes[key] = _SinkProblem(s_off, s_stride, C, p_off, p_stride, rg, out_kn, M, scale, S.dtype,
                                                   tuple(targets))
        self.items.append((S, P, prob))
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        items, self.items, self.armed = self.items, [], False
        if not items:
            return
        probs = tuple([it[2] for it in items])
        plan = self.plans.get(probs)
        if plan is None or plan.partials is not self.partials:
            need = _SinkPlan.elements(probs)
            if self.partials is None or self.partials.shape[1] < need:
                self.plans.clear()  # (they point into the buffer this one replaces)
                self.partials = torch.empty((nat.GRAD_MAX_BLOCKS, need), dtype=torch.float32, device=self.device)
            if len(self.plans) >= 8:
                self.plans.clear()
            plan = self.plans[probs] = _SinkPlan(probs, self.partials, self.device)
        plan.run([it[0].data_ptr() for it in items], [it[1].data_ptr() for it in items])


class _SinkProblem:
    """What a deferred factor-gradient problem is apart from the addresses of its two operands."""
    __slots__ = ("s_off", "s_stride", "C", "p_off", "p_stride", "rg", "out_kn", "M", "scale", "dtype", "targets")

    def __init__(self, *fields):
        for name, value in zip(self.__slots__, fields):
            setattr(self, name, value)


class _SinkPlan:
    """Everything `_AutoSink.flush` needs for one recurring backward pass, built once: the `lora_grad_problem` arrays (one per
    operand dtype) with all fields filled in but the operand pointers, where each Parameter's gradient lies in the flat
    result — Parameters of one shape side by side, so that a step hands its views over with one `view().unbind()` per shape
    instead of a slice and a view per Parameter —, and the device table of the fold.  Per step: two pointer columns written
    into the arrays (numpy, no Python loop over struct fields), the launches, one allocation for the result."""

    @staticmethod
    def elements(probs) -> int:
        return (sum(len(q.targets) * q.rg * q.C for q in probs) + 3) // 4 * 4

    def __init__(self, probs, partials, device):
        import ctypes

        import numpy as np

        self.partials, self.device = partials, device
        self.pstride = partials.shape[1]
        self.stride = self.elements(probs)
        # layout of the flat gradient: targets grouped by (shape, hand-over dtype), in first-seen order
        slots = {}
        for k, q in enumerate(probs):  # (a layer called twice in one pass defers the SAME object twice: slots go by position)
            shape = (q.rg, q.C) if q.out_kn else (q.C, q.rg)
            for i, (p, dt) in enumerate(q.targets):
                slots.setdefault((shape, dt), []).append((k, i, p))
        self.groups, where, off = [], {}, 0
        for (shape, dt), members in slots.items():
            n = shape[0] * shape[1]
            self.groups.append((off, len(members), shape, dt, [p for _, _, p in members]))
            for k, i, _ in members:
                where[(k, i)] = off
                off += n
        assert off <= self.stride
        base = partials.data_ptr()
        rows, by_dtype = [], {}
        for k, q in enumerate(probs):
            by_dtype.setdefault(q.dtype, []).append(k)
            for i in range(len(q.targets)):
                rows.append([where[(k, i)], q.rg * q.C, nat.grad_row_blocks(q.M), 0])
        self.launches = []
        size = ctypes.sizeof(nat.GradProblem)
        for dt, idx in by_dtype.items():
            arr = (nat.GradProblem * len(idx))()
            esize = torch.empty((), dtype=dt).element_size()
            for slot, k in enumerate(idx):
                q, g = probs[k], arr[slot]
                for i in range(len(q.targets)):
                    g.out[i] = base + 4 * where[(k, i)]
                g.s_stride, g.p_stride, g.part_stride, g.M = q.s_stride, q.p_stride, self.pstride, q.M
                g.C, g.r, g.rg, g.out_kn, g.n_blocks, g.scale = q.C, len(q.targets) * q.rg, q.rg, int(q.out_kn), 0, q.scale
            raw = np.frombuffer(arr, dtype=np.uint8)
            s_col = np.ndarray((len(idx),), dtype=np.uint64, buffer=raw, offset=nat.GradProblem.S.offset, strides=(size,))
            p_col = np.ndarray((len(idx),), dtype=np.uint64, buffer=raw, offset=nat.GradProblem.P.offset, strides=(size,))
            s_add = np.array([probs[k].s_off * esize for k in idx], dtype=np.uint64)
            p_add = np.array([probs[k].p_off * 4 for k in idx], dtype=np.uint64)
            self.launches.append((dt, arr, len(idx), np.array(idx, dtype=np.intp), s_col, p_col, s_add, p_add))
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.n_rows, self.max_len = len(rows), max(r_[1] for r_ in rows)
        self.np = np

    def run(self, s_ptrs, p_ptrs):
        np = self.np
        s_all, p_all = np.array(s_ptrs, dtype=np.uint64), np.array(p_ptrs, dtype=np.uint64)
        for dt, arr, n, idx, s_col, p_col, s_add, p_add in self.launches:
            s_col[:] = s_all[idx] + s_add
            p_col[:] = p_all[idx] + p_add
            nat.lora_grad_batched_array(arr, n, dt, self.device)
        grads = torch.empty(self.stride, dtype=torch.float32, device=self.device)
        nat.lora_fold_partials(self.table, self.n_rows, self.max_len, self.partials, self.pstride, grads, False)
        for off, n, shape, dt, params in self.groups:
            views = grads[off:off + n * shape[0] * shape[1]].view(n, *shape)
            if dt != torch.float32:
                views = views.to(dt)
            for p, g in zip(params, views.unbind(0)):
                if not p.requires_grad:
                    continue  # (a factor frozen by the caller: autograd would not have produced its gradient either)
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
        d, u = factor_weights(m)
        return (d.data_ptr(), d._version, u.data_ptr(), u._version)

    def _current(self, members, cdtype, device) -> bool:
        st = self.state
        return (st is not None and self.table is not None and st[0] == cdtype and st[1] == device and
                all(st[2][self.index[id(m)]] == self._sig(m) for m in members))

    def get(self, module, cdtype, d=None, u=None):
        i = self.index.get(id(module))
        if i is None:
            return None
        if d is None:
            d, u = factor_weights(module)
        if d.dtype != torch.float32 or u.dtype != torch.float32 or not d.is_cuda:
            return None  # (factors held in another dtype are cast per call, as before)
        st = self.state
        if (st is None or self.table is None or st[0] != cdtype or st[1] != d.device or
                st[2][i] != (d.data_ptr(), d._version, u.data_ptr(), u._version)) and not self._repack(cdtype, d.device):
            return None
        return self.views[i]

    def ensure(self, members, cdtype) -> bool:
        """True when the packed operands (a group's included) are current for `members` — refreshing them if needed."""
        d = factor_weights(members[0])[0]
        if d.dtype != torch.float32 or not d.is_cuda:
            return False
        return self._current(members, cdtype, d.device) or self._repack(cdtype, d.device)

    def _repack(self, cdtype, device):
        from .groups import QKVGroup, bind_ctx_views, bind_qkv_views, ctx_pack_rows, qkv_pack_rows

        mods = self.modules
        facs = [factor_weights(m) for m in mods]  # (once per optimizer step, for every layer of the model: kept cheap)
        for d, u in facs:
            if d.dtype != torch.float32 or d.device != device or not d.is_contiguous() or not u.is_contiguous():
                return False
        if torch.cuda.is_current_stream_capturing():
            return False
        sigs = tuple([(d.data_ptr(), d._version, u.data_ptr(), u._version) for d, u in facs])
        ptrs = tuple([(sg[0], sg[2]) for sg in sigs])
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
        self.state = (cdtype, device, sigs)
        return True


def register_pack_group(modules) -> None:
    """Called by `inject_trainable_lora` with the modules it wrapped (core.py)."""
    if os.environ.get("DFA_PACK_REGISTRY", "1") == "0":
        return
    modules = list(modules)
    if len(modules) < 2:
        return
    reg = PackRegistry(modules)
    for m in reg.modules:
        m.__dict__["_dfa_packreg"] = reg


_auto_sinks = {}


def deferral_open() -> bool:
    """The process-wide half of `_auto_sink_for`: the switch is on and no process group is alive."""
    return os.environ.get("DFA_DEFER_GRADS", "1") != "0" and not (dist.is_available() and dist.is_initialized())


def param_defers(p) -> bool:
    """The per-Parameter half: a leaf Parameter nobody hooked (a hook wants the gradient autograd would hand it)."""
    return (isinstance(p, torch.nn.Parameter) and p.is_leaf and not p._backward_hooks and
            not getattr(p, "_post_accumulate_grad_hooks", None))


def _auto_sink_for(down, up):
    """The drop-in sink of the tensors' device, or None when deferring is not safe for these Parameters (see _AutoSink)."""
    if not (deferral_open() and param_defers(down) and param_defers(up)):
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


class _FeedForwardFn(torch.autograd.Function):
    """z = (h·gelu(g))·W2ᵀ + b2 with [h | g] = x·Wᵀ + b + s·(x·Aᵀ)·Bᵀ — diffusers FeedForward with a GEGLU activation,
    `net[2](net[0](x))`: the `proj` LoraInjectedLinear with the gate in its forward epilogue (ONE launch writes y = [h | g] and
    the gated activation), the frozen second linear layer on the gated activation, and in backward ONE launch for
    dY = gate-backward(dz·W2, y) (`geglu_linear_bwd`: the gate's backward sits in the epilogue of the linear layer's
    backward-input GEMM) followed by the usual LoRA backward.  One autograd node for the whole block (through round 5 the two
    halves were a node each; the host-bound unchanged-trainer route pays ≈ 25 µs per node)."""

    @staticmethod
    def forward(ctx, x, down, up, w, wt, bias, scale, grad_sink, packed, w2, w2t, b2):
        K, N = w.shape[1], w.shape[0]
        x2 = x.reshape(-1, K)
        if x2.dtype != w.dtype:
            x2 = x2.to(w.dtype)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        a = _as_f32(down)
        b = _as_f32(up)
        packs = packed if packed is not None else nat.lora_pack_factors(a, b, w.dtype)
        res = nat.lora_linear_geglu_fwd(x2, w, bias, a.shape[0], scale, packs, True)
        if res is None:  # no fused kernel for this shape / dtype: the two launches it stands for
            y2, t = nat.lora_linear_fwd(x2, w, bias, a, b, scale, packs)
            gated = nat.geglu_gate_fwd(y2)
        else:
            gated, y2, t = res
        ctx.save_for_backward(x2, a, b, t, y2, w2, w2t)
        ctx.packs = packs
        ctx.wt = wt
        ctx.scale = float(scale)
        ctx.x_shape = x.shape
        ctx.x_dtype = x.dtype
        ctx.factor_dtypes = (down.dtype, up.dtype)
        ctx.grad_sink = grad_sink
        ctx.auto_sink = _auto_sink_for(down, up) if grad_sink is None else None
        ctx.params = (down, up) if ctx.auto_sink is not None else None
        return torch.nn.functional.linear(gated.view(*x.shape[:-1], N // 2), w2, b2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        x2, a, b, t, y2, w2, w2t = ctx.saved_tensors
        dz2 = dz.reshape(-1, dz.shape[-1])
        if dz2.dtype != y2.dtype:
            dz2 = dz2.to(y2.dtype)
        if not dz2.is_contiguous():
            dz2 = dz2.contiguous()
        dy2 = nat.geglu_linear_bwd(dz2, w2t, y2)
        if dy2 is None:  # no fused kernel for this shape / dtype: the two steps it stands for
            dy2 = nat.geglu_gate_bwd(y2, dz2 @ w2)
        return (*_lora_backward(ctx, x2, a, b, t, dy2), None, None, None, None, None, None, None, None, None)


def _frozen_linear(lin, cdtype: torch.dtype):
    """(W, Wᵀ, bias) of a frozen nn.Linear in `cdtype`, cached on the module like _frozen_operands."""
    w, b = linear_params(lin)
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
    return lora_linear(proj_module, x, gate=lin2)


def lora_linear(module, x: torch.Tensor, gate=False) -> torch.Tensor:
    """Fused LoraInjectedLinear forward (lora_diffusion/lora.py:49-50) on the HIP device.  `gate`: False — the plain layer;
    True — with the GEGLU gate behind it (lora_linear_geglu); the frozen nn.Linear that follows the gate — the whole
    feed-forward block (feed_forward_geglu)."""
    global _warned_trainable_base
    w_param, b_param = frozen_linear(module)
    down, up = factor_weights(module)
    if not x.is_cuda or not w_param.is_cuda:
        raise RuntimeError(
            "LoraInjectedLinear.forward: the fused LoRA path runs only on a HIP device (MI355X); got input on "
            f"{x.device} and weight on {w_param.device}. Move the model and inputs to 'cuda' — there is no CPU fallback."
        )
    grad_on = torch.is_grad_enabled()
    if w_param.requires_grad and grad_on and not _warned_trainable_base:
        _warned_trainable_base = True
        warnings.warn(
            "LoraInjectedLinear: the base weight has requires_grad=True, but this path treats W and b as frozen "
            "(no ∇W/∇b are produced). Call model.requires_grad_(False) before inject_trainable_lora as the "
            "reference trainers do."
        )
    cdtype = _compute_dtype(w_param)
    shared = module.__dict__.get("_dfa_shared")  # (group, member): projections of one input called one by one (CLIP q/k/v)
    if shared is not None and not gate and shared[0].usable(x, cdtype) and x.dim() >= 2:
        from .groups import shared_projection

        return shared_projection(shared[0], shared[1], x, cdtype)
    operands = _layer_operands(module, cdtype, grad_on and x.requires_grad, w_param, b_param, down, up)
    if gate is False or gate is True:
        return (_LoraGegluFn if gate else _LoraLinearFn).apply(x, *operands)
    # `gate` is the frozen nn.Linear that follows the gate (feed_forward_geglu): the whole block as one node
    return _FeedForwardFn.apply(x, *operands, *_frozen_linear(gate, cdtype))


def _layer_operands(module, cdtype, need_wt: bool, w_param, b_param, down, up):
    """(down, up, W, Wᵀ | None, bias, scale, grad_sink, packed factors | None) of one wrapped layer: the argument list of the
    LoRA autograd nodes behind the input."""
    w, wt, bias = _frozen_operands(module, cdtype, need_wt, w_param, b_param)
    attrs = module.__dict__
    sink = attrs.get("_dfa_grad_sink")
    packed = attrs.get("_dfa_packed")
    if packed is not None and packed[0].dtype != cdtype:
        packed = None
    if packed is None and sink is None:
        reg = attrs.get("_dfa_packreg")
        if reg is not None:
            packed = reg.get(module, cdtype, down, up)  # drop-in mode: all layers' packed factors from one launch per update
    return down, up, w, wt, bias, float(module.scale), sink, packed


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
