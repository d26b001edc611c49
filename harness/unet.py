"""Build-owned Stable-Diffusion-shaped UNet: the CALLER of the hot path in the benchmark and parity harness.

`diffusers` is not available offline, so the harness needs its own conditional UNet whose LoRA targets enumerate
exactly like diffusers' `UNet2DConditionModel` does for the reference:
  * class NAMES `CrossAttention` and `GEGLU` (matched by name in lora_diffusion/lora.py:53,93-97);
  * per transformer block the registration order attn1, ff, attn2 → to_q, to_k, to_v, to_out.0, ff.net.0.proj,
    to_q, to_k, to_v, to_out.0;
  * top-level registration order down_blocks, up_blocks, mid_block (pinned by the 144-entry index table of
    example_loras/lora_disney.safetensors; tests/test_finder_order.py checks it).
Everything that is NOT the hot path (convolutions, GroupNorm, softmax(QKᵀ)V) is stock PyTorch-ROCm.
Weights are random-init: there are no SD checkpoints offline.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    # True = block has transformer layers (CrossAttnDown/UpBlock2D), False = plain resnet block
    down_attention: Tuple[bool, ...] = (True, True, True, False)
    layers_per_block: int = 2
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    norm_groups: int = 32
    linear_projection: bool = False  # SD2.x uses nn.Linear proj_in/proj_out
    name: str = "sd15"


def sd15_config() -> UNetConfig:
    return UNetConfig()


def sd21_768_config() -> UNetConfig:
    return UNetConfig(num_heads=(5, 10, 20, 20), cross_attention_dim=1024, linear_projection=True, name="sd21-768")


def tiny_config(width: int = 32, cross_dim: int = 32, levels: int = 2) -> UNetConfig:
    """Small same-topology model for parity tests (2 levels, 1 layer per block)."""
    chans = tuple(width * (i + 1) for i in range(levels))
    return UNetConfig(block_out_channels=chans, down_attention=tuple([True] * (levels - 1) + [False]),
                      layers_per_block=1, num_heads=tuple([2] * levels), cross_attention_dim=cross_dim,
                      norm_groups=8, name=f"tiny{width}")


@dataclass
class UNetOutput:
    sample: torch.Tensor


class Timesteps(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)  # flip_sin_to_cos


class TimestepEmbedding(nn.Module):
    def __init__(self, dim_in, dim):
        super().__init__()
        self.linear_1 = nn.Linear(dim_in, dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-5)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class CrossAttention(nn.Module):
    """q/k/v/out projections are the LoRA targets; the attention core is stock SDPA until the caller switches the HIP
    cores on with the reference's hook (`set_use_memory_efficient_attention_xformers(unet, True)`)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        context_dim = query_dim if context_dim is None else context_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def forward(self, x, context=None):
        context = x if context is None else context
        q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)
        return self.to_out[1](self.to_out[0](_attention_core(q, k, v, self.heads)))


def _attention_core(q, k, v, heads):
    """softmax(QKᵀ/√d)V per head on [B, N, H·d] tensors, stock PyTorch SDPA kernels.

    On the GPU the SD head sizes 40 and 80 are zero-padded to 64 and 128: the ROCm SDPA kernels are tuned for
    those sizes (MI355X, 4096 tokens, d = 40, forward+backward: 2.73 ms as is, 0.92 ms padded with the
    memory-efficient backend).  Zero columns add nothing to QKᵀ and produce zero output columns, which are dropped
    again; the scale stays 1/√d of the true head size.  The [B,N,H·d] ↔ [B,H,N,D] re-layouts (with the padding)
    are single streaming kernels (diffusion_finetuning_amd.sandwich) instead of generic strided copies.
    Cross-attention does not come through here once the caller has switched the HIP attention core on
    (`set_use_memory_efficient_attention_xformers(unet, True)`, the reference's own hook): csrc/attn_ctx.hip then works
    on the [B,N,H·d] tensors directly."""
    b, n, hd = q.shape
    d = hd // heads
    if not q.is_cuda:
        split = lambda t: t.view(b, t.shape[1], heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(q), split(k), split(v))
        return o.transpose(1, 2).reshape(b, n, hd)
    from diffusion_finetuning_amd.sandwich import merge_heads, split_heads

    D = 64 if d < 64 else (128 if d < 128 else d)
    q4, k4, v4 = split_heads(q, heads, D), split_heads(k, heads, D), split_heads(v, heads, D)
    # set_priority: without it the list only ENABLES backends and PyTorch still tries flash first, whose backward is
    # 1.65x slower than the memory-efficient one at these shapes (1454 vs 880 us fwd+bwd, 4x8 heads, 4096 tokens)
    with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH], set_priority=True):
        o = F.scaled_dot_product_attention(q4, k4, v4, scale=d ** -0.5)
    return merge_heads(o, d)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):  # the caller's stock composite; `set_use_hip_geglu(model)` swaps in the fused pass
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, heads, dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, context_dim, groups, linear_projection):
        super().__init__()
        self.linear_projection = linear_projection
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Linear(channels, channels) if linear_projection else nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, channels // heads, context_dim)]
        )
        self.proj_out = nn.Linear(channels, channels) if linear_projection else nn.Conv2d(channels, channels, 1)

    def forward(self, x, context):
        b, c, h, w = x.shape
        res = x
        x = self.norm(x)
        if x.is_cuda and not self.linear_projection:
            # A 1×1 convolution IS a linear layer over the channels.  On the GPU run it on the token layout the
            # transformer blocks need anyway (one NCHW→NHWC copy in, one back): the convolution library would
            # transpose to NHWC and back around each of the two convolutions, and handing it the permuted view
            # leaves channels-last tensors behind that every later norm / add / convolution has to re-lay out.
            x = F.linear(x.permute(0, 2, 3, 1).reshape(b, h * w, c), self.proj_in.weight.view(c, c), self.proj_in.bias)
            for blk in self.transformer_blocks:
                x = blk(x, context)
            x = F.linear(x, self.proj_out.weight.view(c, c), self.proj_out.bias)
            return x.view(b, h, w, c).permute(0, 3, 1, 2).contiguous() + res
        if self.linear_projection:
            x = self.proj_in(x.permute(0, 2, 3, 1).reshape(b, h * w, c))
        else:
            x = self.proj_in(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
        for blk in self.transformer_blocks:
            x = blk(x, context)
        if self.linear_projection:
            x = self.proj_out(x).reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
        else:
            x = self.proj_out(x.reshape(b, h, w, c).permute(0, 3, 1, 2))
        return x + res


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, layers, attn, heads, ctx, groups, linear_proj, add_down):
        super().__init__()
        self.attentions = nn.ModuleList(
            [Transformer2DModel(cout, heads, ctx, groups, linear_proj) for _ in range(layers)] if attn else []
        )
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, context):
        outs = []
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if len(self.attentions):
                x = self.attentions[i](x, context)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, c, temb, heads, ctx, groups, linear_proj):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, ctx, groups, linear_proj)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups), ResnetBlock2D(c, c, temb, groups)])

    def forward(self, x, temb, context):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, context)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, layers, attn, heads, ctx, groups, linear_proj, add_up):
        super().__init__()
        self.attentions = nn.ModuleList(
            [Transformer2DModel(cout, heads, ctx, groups, linear_proj) for _ in range(layers)] if attn else []
        )
        resnets = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            inp = cprev if i == 0 else cout
            resnets.append(ResnetBlock2D(inp + skip, cout, temb, groups))
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips, temb, context):
        for i, res in enumerate(self.resnets):
            x = res(torch.cat([x, skips.pop()], dim=1), temb)
            if len(self.attentions):
                x = self.attentions[i](x, context)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNet2DConditionModel(nn.Module):
    """`unet(noisy_latents, timesteps, encoder_hidden_states).sample` like the diffusers model the reference
    trainers call (training_scripts/train_lora_dreambooth.py:843)."""

    def __init__(self, cfg: Optional[UNetConfig] = None):
        super().__init__()
        cfg = cfg or sd15_config()
        self.config = cfg
        ch = cfg.block_out_channels
        temb = ch[0] * 4
        g, ctx, lin = cfg.norm_groups, cfg.cross_attention_dim, cfg.linear_projection
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_proj = Timesteps(ch[0])
        self.time_embedding = TimestepEmbedding(ch[0], temb)

        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i, c in enumerate(ch):
            cin, cout = cout, c
            self.down_blocks.append(DownBlock(cin, cout, temb, cfg.layers_per_block, cfg.down_attention[i],
                                              cfg.num_heads[i], ctx, g, lin, add_down=i < len(ch) - 1))

        # up_blocks are registered BEFORE mid_block: this is the enumeration order the reference's files pin
        self.up_blocks = nn.ModuleList()
        rch = tuple(reversed(ch))
        rattn = tuple(reversed(cfg.down_attention))
        rheads = tuple(reversed(cfg.num_heads))
        cout = rch[0]
        for i, c in enumerate(rch):
            cprev, cout = cout, c
            cin = rch[min(i + 1, len(ch) - 1)]
            self.up_blocks.append(UpBlock(cin, cout, cprev, temb, cfg.layers_per_block + 1, rattn[i], rheads[i], ctx,
                                          g, lin, add_up=i < len(ch) - 1))

        self.mid_block = MidBlock(ch[-1], temb, cfg.num_heads[-1], ctx, g, lin)

        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, timesteps, encoder_hidden_states):
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], device=sample.device)
        timesteps = timesteps.reshape(-1).expand(sample.shape[0])
        temb = self.time_embedding(self.time_proj(timesteps).to(sample.dtype))
        ctx = encoder_hidden_states.to(sample.dtype)
        x = self.conv_in(sample)
        skips: List[torch.Tensor] = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, ctx)
            skips += outs
        x = self.mid_block(x, temb, ctx)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, ctx)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return UNetOutput(sample=x)
