"""Autograd fronts of the two ops sandwiched by the hot path in a transformer block:
the GEGLU gate and the head split/merge around the attention core.  HIP device only, like the rest of the path."""
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
        return nat.attn_merge_heads(g if g.is_contiguous() else g.contiguous(), ctx.d), None, None


class _MergeHeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, d):
        ctx.heads, ctx.D = x.shape[1], x.shape[-1]
        return nat.attn_merge_heads(x if x.is_contiguous() else x.contiguous(), d)

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
