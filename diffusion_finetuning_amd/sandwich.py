"""Autograd fronts of the ops sandwiched by the hot path in a transformer block: the GEGLU gate, the
short-context (cross-) and long-context (self-) attention cores, and the head split/merge for callers that keep their
own attention kernel.
HIP device only, like the rest of the path."""
import torch
from torch.autograd.function import once_differentiable

from . import _native as nat


class _GegluGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        y2 = y.reshape(-1, y.shape[-1])
        if not y2.is_contiguous():
            y2 = y2.contiguous()
        ctx.save_for_backward(y2)
        ctx.shape = y.shape
        return nat.geglu_gate_fwd(y2).view(*y.shape[:-1], y.shape[-1] // 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (y2,) = ctx.saved_tensors
        d2 = dout.reshape(-1, dout.shape[-1])
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        return nat.geglu_gate_bwd(y2, d2).view(ctx.shape)


def geglu_gate(y: torch.Tensor) -> torch.Tensor:
    """h · gelu(g) for y = [h | g] along the last dim (exact gelu), one pass forward and one backward."""
    if not y.is_cuda:
        raise RuntimeError("geglu_gate runs only on a HIP device; there is no CPU fallback")
    return _GegluGateFn.apply(y)


class _SplitHeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, D):
        ctx.d = x.shape[-1] // heads
        return nat.attn_split_heads(x if x.is_contiguous() else x.contiguous(), heads, D)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return nat.attn_merge_heads(g, ctx.d), None, None  # strided views are consumed in place


class _MergeHeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, d):
        ctx.heads, ctx.D = x.shape[1], x.shape[-1]
        return nat.attn_merge_heads(x, d)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return nat.attn_split_heads(g if g.is_contiguous() else g.contiguous(), ctx.heads, ctx.D), None


def split_heads(x: torch.Tensor, heads: int, padded_dim: int) -> torch.Tensor:
    """[B, N, H·d] → [B, H, N, padded_dim] with zero padding."""
    return _SplitHeadsFn.apply(x, heads, padded_dim)


def merge_heads(x: torch.Tensor, head_dim: int) -> torch.Tensor:
    """[B, H, N, D] → [B, N, H·head_dim] (padding dropped)."""
    return _MergeHeadsFn.apply(x, head_dim)


class _CtxAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        q, k, v = (t if t.is_contiguous() else t.contiguous() for t in (q, k, v))
        ctx.save_for_backward(q, k, v)
        ctx.heads, ctx.scale = heads, scale
        return nat.attn_ctx_fwd(q, k, v, heads, scale)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, k, v = ctx.saved_tensors
        dq, dk, dv = nat.attn_ctx_bwd(q, k, v, dout if dout.is_contiguous() else dout.contiguous(), ctx.heads,
                                      ctx.scale)
        return dq, dk, dv, None, None


def ctx_attention_supported(q: torch.Tensor, k: torch.Tensor, heads: int) -> bool:
    """True when `ctx_attention` handles these tensors (f16/bf16 on the HIP device, ≤ 128 keys, head dim ≤ 160)."""
    if not q.is_cuda or q.dim() != 3 or k.dim() != 3 or q.shape[-1] % heads:
        return False
    return nat.attn_ctx_supported(q.shape[0], q.shape[1], k.shape[1], heads, q.shape[-1] // heads, q.dtype)


def ctx_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float = None) -> torch.Tensor:
    """softmax(q·kᵀ·scale)·v per head for a short key/value sequence.  q [B,Tq,H·d], k/v [B,Tk,H·d] → [B,Tq,H·d]:
    the layouts the to_q/to_k/to_v linears produce and to_out consumes, no head split/merge."""
    if not q.is_cuda:
        raise RuntimeError("ctx_attention runs only on a HIP device; there is no CPU fallback")
    if scale is None:
        scale = (q.shape[-1] // heads) ** -0.5
    return _CtxAttentionFn.apply(q, k, v, heads, float(scale))


class _FlashAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        q, k, v = (t if t.is_contiguous() else t.contiguous() for t in (q, k, v))
        need = any(ctx.needs_input_grad[:3])
        out, lse = nat.attn_flash_fwd(q, k, v, heads, scale, want_lse=need)
        if need:
            ctx.save_for_backward(q, k, v, out, lse)
        ctx.heads, ctx.scale = heads, scale
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = nat.attn_flash_bwd(q, k, v, out, dout if dout.is_contiguous() else dout.contiguous(), lse,
                                        ctx.heads, ctx.scale)
        return dq, dk, dv, None, None


def flash_attention_supported(q: torch.Tensor, k: torch.Tensor, heads: int) -> bool:
    """True when `flash_attention` handles these tensors (f16/bf16 on the HIP device, head dim ≤ 160, any lengths)."""
    if not q.is_cuda or q.dim() != 3 or k.dim() != 3 or q.shape[-1] % heads:
        return False
    return nat.attn_flash_supported(q.shape[0], q.shape[1], k.shape[1], heads, q.shape[-1] // heads, q.dtype)


def flash_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float = None) -> torch.Tensor:
    """softmax(q·kᵀ·scale)·v per head for key/value sequences of any length (online softmax over key tiles; nothing of
    size Tq×Tk is ever stored).  Same [B,T,H·d] layouts as `ctx_attention`, which is the faster choice up to 128 keys."""
    if not q.is_cuda:
        raise RuntimeError("flash_attention runs only on a HIP device; there is no CPU fallback")
    if scale is None:
        scale = (q.shape[-1] // heads) ** -0.5
    return _FlashAttentionFn.apply(q, k, v, heads, float(scale))
