"""Grouped LoRA projections: several `LoraInjectedLinear` modules that multiply the SAME input run as one launch.

The reference calls every wrapped linear on its own (diffusers `CrossAttention.forward`: `self.to_q(x)`, `self.to_k(c)`,
`self.to_v(c)`; lora_diffusion/lora.py:49-50 each).  Two groupings are free of any change in arithmetic:

  * `QKVGroup`  — attn1 `to_q / to_k / to_v` of one transformer block share `x`: one fused forward on the concatenated
    frozen weight `[Wq; Wk; Wv]` with a rank-3r block-diagonal factor, one fused dX on its transpose (no `dXq + dXk + dXv`
    accumulation), and the attention core reads / writes the q | k | v column slices of ONE buffer in place.
  * `CtxKVGroup` — attn2 `to_k / to_v` of ALL transformer blocks share `encoder_hidden_states`: one forward launch for the
    32 projections of an SD UNet at the first cross-attention of a pass, one P-only backward launch at the end (the text
    encoder output is frozen: no dX), the cross-attention cores read K / V (and write dK / dV) as column slices.

Groups exist only under a `trainer.LoraSlab` (it owns the packed factors the fused kernels stream and collects the
factor-gradient problems); without one every module keeps running on its own through `ops.lora_linear`.
HIP device only, like the rest of the path.
"""
from typing import List, Optional, Sequence

import torch
from torch.autograd.function import once_differentiable

from . import _native as nat
from ._fastattr import factor_weights, frozen_linear

RANK_PAD = 16  # rank slots of one packed factor (kRP in csrc/lora_gemm.hip)


def _cast_cat(weights: Sequence[torch.Tensor], cdtype: torch.dtype, transpose: bool) -> torch.Tensor:
    parts = []
    for w in weights:
        wd = w.detach()
        wd = wd if wd.is_contiguous() else wd.contiguous()
        parts.append(wd if wd.dtype == cdtype else nat.lora_cast_matrix(wd, cdtype, False))
    cat = torch.cat(parts, dim=0)
    return nat.lora_cast_matrix(cat, cdtype, True) if transpose else cat


class _FrozenCat:
    """[W0; W1; ...] in the compute dtype (and its transpose), rebuilt when any member weight changes."""

    def __init__(self, layers):
        self.layers = layers
        self._key = None
        self._w = self._wt = None

    def get(self, cdtype: torch.dtype, need_wt: bool):
        ws = [frozen_linear(l)[0] for l in self.layers]
        key = tuple((w.data_ptr(), w._version, w.dtype, w.device) for w in ws) + (cdtype,)
        if key != self._key:
            self._key, self._w, self._wt = key, _cast_cat(ws, cdtype, False), None
        if need_wt and self._wt is None:
            self._wt = nat.lora_cast_matrix(self._w, cdtype, True)
        return self._w, self._wt

    def bias(self, cdtype: torch.dtype):
        """[b0; b1; ...] in the compute dtype (None when the members have no bias), rebuilt when a member's bias changes."""
        bs = [frozen_linear(l)[1] for l in self.layers]
        if bs[0] is None:
            return None
        key = tuple((b.data_ptr(), b._version, b.dtype, b.device) for b in bs) + (cdtype,)
        if key != getattr(self, "_bkey", None):
            self._bkey, self._b = key, torch.cat([b.detach().to(cdtype) for b in bs]).contiguous()
        return self._b


_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)


_top_saved_hooks = getattr(torch._C._autograd, "_top_saved_tensors_default_hooks", None)


def _in_backward() -> bool:
    return _graph_task_id is not None and _graph_task_id() != -1


def _saved_tensor_hooks_active() -> bool:
    """True inside torch.utils.checkpoint(use_reentrant=False) (and save_on_cpu): what a function saves there is
    recomputed / moved per region, so a buffer shared by several regions must not be created inside one."""
    return _top_saved_hooks is not None and _top_saved_hooks(False) is not None


def _same_scale(layers) -> Optional[float]:
    s = float(layers[0].scale)
    return s if all(float(l.scale) == s for l in layers) else None


def qkv_pack_rows(grp, src_offs, base: int):
    """Rows of the one-launch re-pack (`lora_pack_items`: {src_off, which, len, r, d16_off, d16_ld, dT_off, rows}) that fill a
    QKVGroup's packed operands at element offset `base` of a packed buffer; src_offs[g] = (up_off, down_off) of member g's
    fp32 factors relative to the launch's `params` pointer.  Returns (rows, (fa, qb, fb, qa), elements used)."""
    K, N, r, G = grp.K, grp.N, grp.r, grp.G
    rows = []
    if grp.wide:  # every member keeps its own 16-slot factors: Fa [G][16,K] | Qb [GN,16] | Fb [16,GN] | Qa [G][K,16]
        fa, qb, fb, qa = base, base + 16 * G * K, base + 16 * G * K + 16 * G * N, base + 16 * G * K + 32 * G * N
        for g, (up_off, down_off) in enumerate(src_offs):
            rows.append([down_off, 0, K, r, fa + g * 16 * K, K, qa + g * 16 * K, 16])
            rows.append([up_off, 1, N, r, fb + g * N, G * N, qb + g * N * 16, 16])
        return rows, (fa, qb, fb, qa), 32 * G * (K + N)
    # block-diagonal rank-G·r factors (the buffer must have been ZEROED once: every member fills only its own rank slots)
    fa, qb, fb, qa = base, base + 16 * K, base + 16 * K + 16 * G * N, base + 16 * K + 32 * G * N
    for g, (up_off, down_off) in enumerate(src_offs):
        rows.append([down_off, 0, K, r, fa + g * r * K, K, qa + g * r, r])
        rows.append([up_off, 1, N, r, fb + g * r * G * N + g * N, G * N, qb + g * N * 16 + g * r, r])
    return rows, (fa, qb, fb, qa), 32 * K + 32 * G * N


def bind_qkv_views(grp, pk, spec):
    fa, qb, fb, qa = spec
    GN = grp.G * grp.N
    nk = 16 * grp.K * (grp.G if grp.wide else 1)
    grp.Fa, grp.Qb, grp.Fb, grp.Qa = pk[fa:fa + nk], pk[qb:qb + 16 * GN], pk[fb:fb + 16 * GN], pk[qa:qa + nk]


def ctx_pack_rows(grp, src_offs, base: int):
    """The same for a CtxKVGroup: A16 [G][16,K] | B16 [ΣN,16] | Bt16 [part g: [16, N_g] at 16·off_g]."""
    K, r, G = grp.K, grp.r, grp.G
    a16, b16, bt = base, base + 16 * G * K, base + 16 * G * K + 16 * grp.total
    rows = []
    grp.bt_off = []
    for g, (up_off, down_off) in enumerate(src_offs):
        rows.append([down_off, 0, K, r, a16 + g * 16 * K, K, -1, 16])
        rows.append([up_off, 1, grp.N[g], r, bt + 16 * grp.off[g], grp.N[g], b16 + 16 * grp.off[g], 16])
        grp.bt_off.append(16 * grp.off[g])
    return rows, (a16, b16, bt), 16 * G * K + 32 * grp.total


def bind_ctx_views(grp, pk, spec):
    a16, b16, bt = spec
    grp.A16 = pk[a16:a16 + 16 * grp.G * grp.K]
    grp.B16 = pk[b16:b16 + 16 * grp.total]
    grp.Bt16 = pk[bt:bt + 16 * grp.total]


class _GradTargets:
    """Where a group's factor-gradient problems go.  Under a trainer: the slab's batched launch, outputs = the members' slots
    of the partial-sum slab.  Without one (an unchanged reference trainer calling loss.backward()): the drop-in sink, outputs =
    the members' Parameters (ops._AutoSink hands them their .grad when the backward pass ends)."""

    def __init__(self, group):
        self.g = group

    def note(self, M, need_dx):
        if self.g.sinks is not None:
            slab = self.g.sinks[0].slab
            for sink in self.g.sinks:
                slab.note_layer(sink.index, M, need_dx)

    def defer(self, members, up: bool, S, s_off, s_stride, C, P, p_off, p_stride, M, scale, keep):
        g = self.g
        r = g.r
        if g.sinks is not None:
            slab = g.sinks[0].slab
            outs = [(g.sinks[i].up_ptr if up else g.sinks[i].down_ptr) for i in members]
            slab.defer(nat.grad_problem(S, s_off, s_stride, C, P, p_off, p_stride, len(members) * r, outs, r, not up,
                                        slab.stride, M, scale), g.sinks[members[0]].index if up else None, keep)
        else:
            from .ops import _auto_sink_for

            sink = _auto_sink_for(*factor_weights(g.layers[members[0]]))
            targets = [(factor_weights(g.layers[i])[1 if up else 0], torch.float32) for i in members]
            sink.defer_problem(S, s_off, s_stride, C, P, p_off, p_stride, r, not up, M, scale, targets)


def _dropin_ready(group, cdtype) -> bool:
    """A group without a slab: its packed operands come from the members' ops.PackRegistry (refreshed here when a factor
    changed) and its gradients go to the drop-in sink — which must exist for every member (no process group, no hooks)."""
    from .ops import deferral_open, param_defers

    reg = group.registry
    if reg is None or not deferral_open():
        return False
    for l in group.layers:
        if "_dfa_grad_sink" in l.__dict__:
            return False  # the member joined a trainer's slab since: its gradients belong there (the slab's own groups, if any)
        down, up = factor_weights(l)
        if not (param_defers(down) and param_defers(up)):
            return False
    return reg.ensure(group.layers, cdtype)


def _members_frozen(layers) -> bool:
    """No member's frozen weight asks for a gradient (the grouped launches produce none)."""
    for l in layers:
        if frozen_linear(l)[0].requires_grad:
            return False
    return True


class QKVGroup:
    """to_q / to_k / to_v of one self-attention module.  Packed operands (views into the slab's packed buffer):
         Fa [16,K]   rows g·r+j = A_g[j,:]           (forward main-loop factor)
         Qb [3N,16]  row g·N+n, col g·r+j = B_g[n,j]  (forward epilogue factor, block diagonal)
         Fb [16,3N]  row g·r+j, col g·N+n = B_g[n,j]  (backward main-loop factor, block diagonal)
         Qa [K,16]   row k, col g·r+j = A_g[j,k]      (backward epilogue factor)
    `wide` (3r > 16 rank slots — the ranks of BASELINE configs 3 and 5): every member keeps its own 16-slot factor pair and
    the launch is `lora_gemm_parts` (column runs forward, contraction runs backward):
         Fa [3][16,K]  part g: rows j < r = A_g        Qb [3N,16]    row g·N+n, col j < r = B_g[n,j]
         Fb [16,3N]    row j < r, col g·N+n = B_g[n,j]  Qa [3][K,16]  part g: row k, col j < r = A_g[j,k]
    T / U keep the [M, 3r] layout (member g in columns g·r ..) either way."""

    def __init__(self, layers, sinks):
        self.layers, self.sinks = list(layers), (None if sinks is None else list(sinks))
        lin = layers[0].linear
        self.K, self.N = lin.in_features, lin.out_features
        self.r = layers[0].lora_down.weight.shape[0]
        self.G = len(layers)
        self.wide = self.G * self.r > RANK_PAD
        self.frozen = _FrozenCat(self.layers)
        self.Fa = self.Qb = self.Fb = self.Qa = None  # set by LoraSlab.enable_packed (or by the members' PackRegistry)
        self.registry = None                          # drop-in mode (sinks is None): ops.PackRegistry that owns the operands
        self.Fb_part = None
        self.grads = _GradTargets(self)

    @staticmethod
    def eligible(layers) -> bool:
        from .core import LoraInjectedLinear

        if not all(isinstance(l, LoraInjectedLinear) for l in layers):
            return False
        lin = layers[0].linear
        r = layers[0].lora_down.weight.shape[0]
        # (biases: all members or none — the concatenated bias rides on the kernel's own bias argument; CLIP's projections
        #  carry one, the UNet's q/k/v do not)
        fits = len(layers) * r <= RANK_PAD or (r <= RANK_PAD and len(layers) == 3)  # (the part-wise backward: three members)
        return (fits and lin.in_features % 64 == 0 and lin.out_features % 64 == 0 and
                all(l.linear.in_features == lin.in_features and l.linear.out_features == lin.out_features and
                    (l.linear.bias is None) == (lin.bias is None) and l.lora_down.weight.shape[0] == r for l in layers))

    def usable(self, x: torch.Tensor, cdtype: torch.dtype) -> bool:
        if self.wide and cdtype == torch.float32:
            return False  # the part-wise kernels are 16-bit (fp32 parity runs keep the members on their own)
        if self.sinks is None and not (x.is_cuda and _dropin_ready(self, cdtype)):
            return False
        return (self.Fa is not None and self.Fa.dtype == cdtype and x.is_cuda and _same_scale(self.layers) is not None
                and _members_frozen(self.layers))


class _QKVProjFn(torch.autograd.Function):
    """qkv[M, 3N] = x·[Wq;Wk;Wv]ᵀ + s·(x·A_catᵀ)·B_bdᵀ — three lora.py:49-50 forwards in one launch; backward = one fused
    dX launch; the six factor gradients are handed to the slab's batched gradient launch."""

    @staticmethod
    def forward(ctx, x, group, cdtype, *factors):  # factors: the (down, up) Parameters, listed so autograd tracks them
        x2, t, qkv = _qkv_project(ctx, x, group, cdtype)
        ctx.save_for_backward(x2, t)
        return qkv.view(*x.shape[:-1], qkv.shape[1])

    @staticmethod
    @once_differentiable
    def backward(ctx, dqkv):
        x2, t = ctx.saved_tensors
        return _qkv_project_backward(ctx, x2, t, dqkv)


def _qkv_project(ctx, x, group, cdtype):
    """Forward body of a grouped q/k/v projection for the autograd node `ctx` (which saves what it needs itself): returns
    (x2 [M,K], T [M,G·r] fp32, qkv [M,G·N])."""
    K, N3, rr = group.K, group.G * group.N, group.G * group.r
    x2 = x.reshape(-1, K)
    if x2.dtype != cdtype:
        x2 = x2.to(cdtype)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    w, wt = group.frozen.get(cdtype, ctx.needs_input_grad[0])  # Wᵀ only when a dX launch will follow
    M = x2.shape[0]
    qkv = torch.empty((M, N3), dtype=cdtype, device=x2.device)
    t = torch.empty((M, rr), dtype=torch.float32, device=x2.device)
    scale = _same_scale(group.layers)
    if group.wide:
        if not nat.lora_gemm_parts(x2, w, group.frozen.bias(cdtype), group.Fa, group.Qb, qkv, t, rr, M, K, N3, group.r,
                                   group.G, False, scale):
            raise RuntimeError("grouped q/k/v forward: lora_gemm_parts has no kernel for this shape")
    else:
        nat.lora_gemm_packed(x2, K, w, group.frozen.bias(cdtype), group.Fa, group.Qb, None, None, 0, qkv, t, M, K, N3, rr,
                             scale)
    ctx.group, ctx.wt, ctx.scale = group, wt, scale
    ctx.x_shape, ctx.x_dtype = x.shape, x.dtype
    return x2, t, qkv


def _qkv_project_backward(ctx, x2, t, dqkv):
    """Backward body: one fused dX launch (when the input wants a gradient), the factor gradients deferred; returns the
    node's gradient tuple for (x, group, cdtype, *factors)."""
    g = ctx.group
    K, N, N3, r, rr = g.K, g.N, g.G * g.N, g.r, g.G * g.r
    d2 = dqkv.reshape(-1, N3)
    if d2.dtype != x2.dtype:
        d2 = d2.to(x2.dtype)
    if not d2.is_contiguous():
        d2 = d2.contiguous()
    M = d2.shape[0]
    need_dx = ctx.needs_input_grad[0]
    u = torch.empty((M, rr), dtype=torch.float32, device=d2.device)
    u_by_part = False  # U layout: [M, 3r] (member g in columns g·r ..) or, from per-member launches, [3][M, r]
    dx2 = None
    if need_dx:
        if ctx.wt is None:
            raise RuntimeError("grouped q/k/v backward: Wᵀ operand was not prepared in forward")
        dx2 = torch.empty((M, K), dtype=d2.dtype, device=d2.device)
        if g.wide:
            if not nat.lora_gemm_parts(d2, ctx.wt, None, g.Fb, g.Qa, dx2, u, rr, M, N3, K, r, g.G, True, ctx.scale):
                raise RuntimeError("grouped q/k/v backward: lora_gemm_parts has no kernel for this shape")
        else:
            nat.lora_gemm_packed(d2, N3, ctx.wt, None, g.Fb, g.Qa, None, None, 0, dx2, u, M, N3, K, rr, ctx.scale)
    elif g.wide:
        # no dX wanted (the first block of a model: its input carries no gradient): U_g = dY_g·B_g from the members'
        # own packed factors, one small launch each on the column slices of the shared gradient buffer
        u, u_by_part = u.view(g.G, M, r), True
        for i in range(g.G):
            nat.lora_gemm_packed(d2[:, i * N:(i + 1) * N], N3, None, None, g.Fb_part[i], None, None, None, 0, None, u[i],
                                 M, N, 0, r, ctx.scale)
    else:
        nat.lora_gemm_packed(d2, N3, None, None, g.Fb, None, None, None, 0, None, u, M, N3, 0, rr, ctx.scale)
    g.grads.note(M, need_dx)
    for i in range(g.G):  # gB_i = s·dY_iᵀ·T_i : column slices of the shared buffers
        g.grads.defer([i], True, d2, i * N, N3, N, t, i * r, rr, M, ctx.scale, (d2, t))
    # gA = s·Uᵀ·X: as many members per problem as fit 16 accumulator columns (rank groups of r → their `down` gradients);
    # all three at rank <= 5 — X read once — two + one at rank 8, one each at rank 16
    per = 1 if u_by_part else max(1, RANK_PAD // r)
    for c in range(0, g.G, per):
        mem = list(range(c, min(c + per, g.G)))
        off, ld = (c * M * r, r) if u_by_part else (c * r, rr)
        g.grads.defer(mem, False, x2, 0, K, K, u, off, ld, M, ctx.scale, (x2, u))
    dx = None
    if need_dx:
        dx = dx2.view(ctx.x_shape)
        if dx.dtype != ctx.x_dtype:
            dx = dx.to(ctx.x_dtype)
    return (dx, None, None) + (None,) * (2 * g.G)


class _QKVAttnFn(torch.autograd.Function):
    """The grouped q/k/v projection (`_QKVProjFn`'s two bodies) and the self-attention core on the q | k | v column slices of
    its [B,T,3·H·d] output as ONE autograd node: backward writes dq | dk | dv as the column slices of one buffer — the dY of the
    projection, no split / cat copies either way.  What a self-attention module's forward uses (`qkv_self_attention`); through
    round 5 projection and core were a node each (≈ 25 µs of host time per node on the unchanged-trainer route)."""

    @staticmethod
    def forward(ctx, x, group, cdtype, heads, attn_scale, tail, *factors):
        # factors: the members' (down, up) Parameters, then — with a `tail` (ops.LoraTail: the module's `to_out[0]`, run here
        # behind the core) — that layer's two
        x2, t, qkv = _qkv_project(ctx, x, group, cdtype)
        qkv = qkv.view(*x.shape[:-1], qkv.shape[1])
        need = any(ctx.needs_input_grad)
        out, lse = nat.attn_flash_fwd_qkv(qkv, heads, attn_scale, want_lse=need)
        ctx.heads, ctx.attn_scale, ctx.tail = heads, attn_scale, tail
        if tail is None:
            if need:
                ctx.save_for_backward(x2, t, qkv, out, lse)
            return out
        y, kept = tail.forward(out, ctx.needs_input_grad[-2], ctx.needs_input_grad[-1])
        if need:
            ctx.save_for_backward(x2, t, qkv, out, lse, *kept)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x2, t, qkv, out, lse, *kept = ctx.saved_tensors
        tail_grads = ()
        if ctx.tail is not None:
            dout, g_down, g_up = ctx.tail.backward(dout, *kept)
            tail_grads = (g_down, g_up)
        dqkv = nat.attn_flash_bwd_qkv(qkv, out, dout if dout.is_contiguous() else dout.contiguous(), lse, ctx.heads,
                                      ctx.attn_scale)
        grads = _qkv_project_backward(ctx, x2, t, dqkv)
        return grads[:3] + (None, None, None) + grads[3:] + tail_grads


class _SplitQKVFn(torch.autograd.Function):
    """[.., 3N] → three [.., N] column-slice VIEWS of it (no copies); backward packs the three incoming gradients into one
    contiguous [.., 3N] buffer — the dY of `_QKVProjFn` — with one concatenation."""

    @staticmethod
    def forward(ctx, qkv, n):
        ctx.n = n
        return tuple(qkv[..., i * n:(i + 1) * n] for i in range(qkv.shape[-1] // n))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        like = next(g for g in grads if g is not None)
        return torch.cat([g if g is not None else torch.zeros_like(like) for g in grads], dim=-1), None


def shared_projection(group: QKVGroup, member: int, x: torch.Tensor, cdtype: torch.dtype) -> torch.Tensor:
    """Output of member `member` of a group whose members are called ONE BY ONE by a forward this package does not own —
    transformers' `CLIPAttention.forward`: `q_proj(h)`, `k_proj(h)`, `v_proj(h)` (LoRA target class "CLIPAttention",
    lora_diffusion/lora.py:54).  The first member called on a tensor launches the grouped projection and keeps the three
    column slices; the other members, called on the SAME tensor, take theirs.  The results are views of one [M, 3N] buffer
    (row stride 3N): what `.view(B, T, heads, d)` and the attention cores accept as they are."""
    memo = group.__dict__.get("_memo")
    key = (id(x), x._version, torch.is_grad_enabled())
    if memo is None or memo[0] != key:
        factors = [p for l in group.layers for p in factor_weights(l)]
        parts = _SplitQKVFn.apply(_QKVProjFn.apply(x, group, cdtype, *factors), group.N)
        memo = group.__dict__["_memo"] = [key, x, list(parts), 0]  # (x itself is kept: its id cannot be reused meanwhile)
    out = memo[2][member]
    memo[3] += 1
    if memo[3] >= group.G:
        group.__dict__["_memo"] = None  # every member has taken its slice: drop the references
    return out


def qkv_self_attention(group: QKVGroup, x: torch.Tensor, heads: int, scale: Optional[float], cdtype: torch.dtype, tail=None):
    """softmax(q kᵀ·scale) v for q, k, v = the group's three LoRA projections of x → [B, T, H·d]; with a `tail` (ops.LoraTail
    of the module's `to_out[0]`) that layer's output instead, from the same autograd node."""
    factors = [p for l in group.layers for p in factor_weights(l)]
    if tail is not None:
        factors += tail.factors
    if scale is None:
        scale = (group.N // heads) ** -0.5
    return _QKVAttnFn.apply(x, group, cdtype, heads, float(scale), tail, *factors)


class _CtxPass:
    """What one forward pass of the model shares between its cross-attention modules."""

    def __init__(self, key, kv):
        self.key, self.kv = key, kv
        self.consumers = 0       # cross-attentions that took their K / V from `kv` in this pass
        self.returned = 0        # ... whose backward has run
        self.dkv = None          # gradient buffer, same layout as kv


class CtxKVGroup:
    """to_k / to_v of every cross-attention module that projects the same encoder_hidden_states.  Part 2i = to_k and
    part 2i+1 = to_v of module i; columns [off[g], off[g] + N[g]) of the [M, ΣN] buffers belong to part g.  Forward: one
    launch per pass.  Backward: one P-only launch for the factors' U = dY·B of every part, and — when the context itself
    carries a gradient (a training text encoder / training token embeddings) — one launch for its dX."""

    def __init__(self, modules, layers, sinks):
        self.modules, self.layers, self.sinks = list(modules), list(layers), (None if sinks is None else list(sinks))
        self.G = len(self.layers)
        self.K = self.layers[0].linear.in_features
        self.r = self.layers[0].lora_down.weight.shape[0]
        self.N = [l.linear.out_features for l in self.layers]
        self.off, o = [], 0
        for n in self.N:
            self.off.append(o)
            o += n
        self.total = o
        self.frozen = _FrozenCat(self.layers)
        self.A16 = self.B16 = self.Bt16 = None  # packed operands: set by LoraSlab.enable_packed (or the members' PackRegistry)
        self.registry = None                     # drop-in mode (sinks is None)
        self.grads = _GradTargets(self)
        self.bt_off: List[int] = []              # element offset of part g's Bt16 [16, N_g] inside self.Bt16
        self.tile_part = None
        self._part_tables = {}
        self._pass: Optional[_CtxPass] = None

    @staticmethod
    def eligible(layers) -> bool:
        from .core import LoraInjectedLinear

        if not layers or not all(isinstance(l, LoraInjectedLinear) for l in layers):
            return False
        k, r = layers[0].linear.in_features, layers[0].lora_down.weight.shape[0]
        return (r <= RANK_PAD and k % 64 == 0 and
                all(l.linear.in_features == k and l.linear.out_features % 64 == 0 and l.linear.bias is None and
                    l.lora_down.weight.shape[0] == r for l in layers))

    def new_pass(self):
        self._pass = None

    def check_pass_complete(self):
        """Called when the slab collects a step's factor gradients: a backward pass that reached SOME of the pass's
        cross-attentions but not all (one whose output took no part in the loss) never handed the shared dK/dV buffer to
        autograd — every to_k/to_v factor gradient of the group would silently be zero."""
        st = self._pass
        if st is not None and 0 < st.returned < st.consumers:
            raise RuntimeError(
                f"CtxKVGroup: backward reached {st.returned} of the {st.consumers} cross-attentions that shared this pass's "
                "K/V projection; the grouped projection's gradient was never produced.  Build the trainer with "
                "group_projections=False for models whose loss does not depend on every cross-attention.")

    def usable(self, ctx_t: torch.Tensor, cdtype: torch.dtype) -> bool:
        # Not under gradient checkpointing (train_lora_dreambooth.py:627-630): it re-runs one block's forward at a time
        # inside backward (reentrant form) or replays what a block saved (non-reentrant form), and the group's shared
        # buffers belong to a whole forward pass, not to a block.  Those cross-attentions project their own K / V
        # (per-module path); the reentrant form's no-grad forward and every ordinary pass use the group.
        if _in_backward() or _saved_tensor_hooks_active():
            return False
        # The members (32 layers in an SD UNet) are checked once per pass — by the cross-attention that opens it, whose
        # projection launch serves all of them: packed operands current and gradients deferrable (drop-in mode), one LoRA
        # scale, frozen weights.  The other 15 consumers of the pass only check what is their own.
        if self._pass is None:
            if self.sinks is None and not (ctx_t.is_cuda and _dropin_ready(self, cdtype)):
                return False
            if _same_scale(self.layers) is None or not _members_frozen(self.layers):
                return False
        # (a context that carries a gradient — a text encoder that trains, train_lora_dreambooth.py:608-621, or whose token
        #  embeddings do, cli_lora_pti.py:706-722 — gets its dX from ONE grouped launch as well: _CtxProjFn.backward)
        return (self.A16 is not None and self.A16.dtype == cdtype and ctx_t.is_cuda and ctx_t.dim() == 3 and
                ctx_t.shape[-1] == self.K)

    def tables(self, device, M: int):
        if self.tile_part is None:
            tp = []
            for g, n in enumerate(self.N):
                tp += [g | (1 << 16)] + [g] * (n // 64 - 1)
            self.tile_part = torch.tensor(tp, dtype=torch.int32).to(device)
        pt = self._part_tables.get(M)
        if pt is None:
            rows = [[self.off[g], self.N[g], self.bt_off[g], g * M * self.r] for g in range(self.G)]
            pt = self._part_tables[M] = torch.tensor(rows, dtype=torch.int64).to(device)
        return self.tile_part, pt

    def project(self, ctx_t: torch.Tensor, cdtype: torch.dtype) -> _CtxPass:
        """The [B·L, ΣN] buffer of all K / V projections of `ctx_t`, computed by the first cross-attention of a pass."""
        key = (ctx_t.data_ptr(), ctx_t._version, tuple(ctx_t.shape), ctx_t.dtype, torch.is_grad_enabled())
        st = self._pass
        if st is None or st.key != key:
            factors = [p for l in self.layers for p in factor_weights(l)]
            st = self._pass = _CtxPass(key, None)
            st.kv = _CtxProjFn.apply(ctx_t, self, cdtype, st, *factors)
        return st


class _CtxProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ehs, group, cdtype, state, *factors):
        K = group.K
        e2 = ehs.reshape(-1, K)
        if e2.dtype != cdtype:
            e2 = e2.to(cdtype)
        if not e2.is_contiguous():
            e2 = e2.contiguous()
        M = e2.shape[0]
        w, _ = group.frozen.get(cdtype, ctx.needs_input_grad[0])  # Wᵀ only when the context wants a gradient
        tile_part, _ = group.tables(e2.device, M)
        kv = torch.empty((M, group.total), dtype=cdtype, device=e2.device)
        t = torch.empty((group.G, M, group.r), dtype=torch.float32, device=e2.device)
        scale = _same_scale(group.layers)
        nat.lora_gemm_packed(e2, K, w, None, group.A16, group.B16, tile_part, None, group.G, kv, t, M, K, group.total,
                             group.r, scale)
        ctx.save_for_backward(e2, t)
        ctx.group, ctx.scale = group, scale  # (not `state`: state → kv → this node would be a reference cycle)
        ctx.e_shape, ctx.e_dtype = ehs.shape, ehs.dtype
        return kv

    @staticmethod
    @once_differentiable
    def backward(ctx, dkv):
        g = ctx.group
        e2, t = ctx.saved_tensors
        M, r = e2.shape[0], g.r
        d2 = dkv if dkv.is_contiguous() else dkv.contiguous()
        _, part_table = g.tables(e2.device, M)
        u = torch.empty((g.G, M, r), dtype=torch.float32, device=e2.device)
        nat.lora_gemm_packed(d2, g.total, None, None, g.Bt16, None, None, part_table, g.G, None, u, M, 64, 0, r,
                             ctx.scale, work_cols=g.total)
        need_dx = ctx.needs_input_grad[0]
        de = None
        if need_dx:
            # d(context) = Σ_g dKV_g·W_g + s·Σ_g U_g·A_g — the dX of all G projections at once, no per-layer launches and no
            # G-1 accumulations: ONE fused-GEMM launch over the concatenated contraction (dKV [M, ΣN] · [W_0; W_1; …], split
            # over K inside the launch: M is only batch × 77 rows) plus the rank-(G·r) product of the U the launch above left
            # ([M, G·r]·[G·r, K], a plain library GEMM on the fp32 masters)
            _, wt = g.frozen.get(d2.dtype, True)
            zeros = nat.zero_factor_buffer(16 * max(g.total, g.K), d2.dtype, d2.device)
            if zeros is None:
                raise RuntimeError("grouped context projection backward: run one host-launched step before recording (the "
                                   "zero-factor buffer is allocated outside a recording)")
            de2 = torch.empty((M, g.K), dtype=d2.dtype, device=d2.device)
            nat.lora_gemm_packed(d2, g.total, wt, None, zeros, zeros, None, None, 0, de2, None, M, g.total, g.K, 1, 0.0)
            acat = torch.cat([l.lora_down.weight.detach().float() for l in g.layers])      # [G·r, K]
            de = torch.addmm(de2.float(), u.permute(1, 0, 2).reshape(M, g.G * r), acat, alpha=ctx.scale)
            de = de.to(ctx.e_dtype).view(ctx.e_shape)
        g.grads.note(M, need_dx)
        for i in range(g.G):
            g.grads.defer([i], True, d2, g.off[i], g.total, g.N[i], t, i * M * r, r, M, ctx.scale, (d2, t))
            g.grads.defer([i], False, e2, 0, g.K, g.K, u, i * M * r, r, M, ctx.scale, (e2, u))
        return (de, None, None, None) + (None,) * (2 * g.G)


class _CtxAttnKVFn(torch.autograd.Function):
    """Cross-attention core of module `index` of a CtxKVGroup: K / V are column slices of the group's buffer; backward
    writes dK / dV into the same slices of the group's gradient buffer.  The buffer is handed to autograd exactly once —
    by the last of the pass's cross-attentions to run backward — so the projections' backward sees every slice filled
    and nothing is ever added or copied."""

    @staticmethod
    def forward(ctx, q, kv, group, state, index, heads, scale, tail, *tail_factors):
        # tail: ops.LoraTail of the module's `to_out[0]` (run here, behind the core; its two factors in tail_factors) or None
        q = q if q.is_contiguous() else q.contiguous()
        off_k, off_v = group.off[2 * index], group.off[2 * index + 1]
        out = nat.attn_ctx_fwd_kv(q, kv, off_k, off_v, heads, scale)
        kept = ()
        if tail is not None:
            out, kept = tail.forward(out, ctx.needs_input_grad[-2], ctx.needs_input_grad[-1])
        ctx.core = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]  # (else only the tail's factors want a gradient)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(q, kv, *kept)
            if ctx.core:
                state.consumers += 1
        ctx.state, ctx.offs, ctx.heads, ctx.scale, ctx.tail = state, (off_k, off_v), heads, scale, tail
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, kv, *kept = ctx.saved_tensors
        tail_grads = ()
        if ctx.tail is not None:
            dout, g_down, g_up = ctx.tail.backward(dout, *kept)
            tail_grads = (g_down, g_up)
        if not ctx.core:
            return (None,) * 8 + tail_grads
        st = ctx.state
        if st.dkv is None:
            st.dkv = torch.zeros_like(kv)
        dq = nat.attn_ctx_bwd_kv(q, kv, st.dkv, ctx.offs[0], ctx.offs[1], dout if dout.is_contiguous() else dout.contiguous(),
                                 ctx.heads, ctx.scale)
        st.returned += 1
        dkv = None
        if st.returned == st.consumers:
            # the last consumer hands the buffer over and closes the pass: a second backward over the same graph
            # (retain_graph) starts counting — and filling a fresh buffer — again
            dkv, st.dkv, st.returned = st.dkv, None, 0
        return (dq, dkv, None, None, None, None, None, None) + tail_grads


def ctx_cross_attention(group: CtxKVGroup, index: int, q: torch.Tensor, ctx_t: torch.Tensor, heads: int,
                        scale: Optional[float], cdtype: torch.dtype, tail=None) -> torch.Tensor:
    """Cross-attention of module `index` of the group on its query projection `q`; with a `tail` (ops.LoraTail of the module's
    `to_out[0]`) that layer's output instead, from the same autograd node."""
    st = group.project(ctx_t, cdtype)
    if scale is None:
        scale = (q.shape[-1] // heads) ** -0.5
    if q.dtype != cdtype:
        q = q.to(cdtype)
    if tail is None:
        return _CtxAttnKVFn.apply(q, st.kv, group, st, index, heads, float(scale), None)
    return _CtxAttnKVFn.apply(q, st.kv, group, st, index, heads, float(scale), tail, *tail.factors)
