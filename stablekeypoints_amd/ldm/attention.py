"""Attention building blocks with the diffusers==0.8.0 module/attribute layout [3P].

The reference's patcher (ptp_utils.py:555-568) finds modules by class NAME `CrossAttention` and
uses `heads`, `scale`, `to_q/to_k/to_v`, `to_out`, `reshape_heads_to_batch_dim`,
`reshape_batch_dim_to_heads` (ptp_utils.py:474-491,540), so those are kept verbatim as an
interface; state-dict keys match the published SD-1.x UNet checkpoints.

`CrossAttention.forward` here is the plain (unpatched) attention the reference inherits from
diffusers for the down/mid blocks.  The hooked up-block path is installed by
`stablekeypoints_amd.ptp_utils.register_attention_control`.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class CrossAttention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim=None, heads: int = 8, dim_head: int = 64):
        super().__init__()
        inner = heads * dim_head
        ctx = query_dim if cross_attention_dim is None else cross_attention_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(ctx, inner, bias=False)
        self.to_v = nn.Linear(ctx, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def reshape_heads_to_batch_dim(self, t):
        b, n, c = t.shape
        h = self.heads
        return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def reshape_batch_dim_to_heads(self, t):
        bh, n, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, d * h)

    def forward(self, hidden_states, context=None, mask=None):
        ctx = hidden_states if context is None else context
        q = self.reshape_heads_to_batch_dim(self.to_q(hidden_states))
        k = self.reshape_heads_to_batch_dim(self.to_k(ctx))
        v = self.reshape_heads_to_batch_dim(self.to_v(ctx))
        attn = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
                             q, k.transpose(1, 2), beta=0, alpha=self.scale).softmax(dim=-1)
        out = self.reshape_batch_dim_to_heads(torch.bmm(attn, v))
        return self.to_out[0](out)


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        p = self.proj(x)
        if p.is_cuda and p.dtype == torch.float32 and p.shape[-1] % 8 == 0:
            from .. import ops
            return ops.geglu(p)                       # one fused pass per direction on the HIP kernel
        h, gate = p.chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: int):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, cross_attention_dim, heads, dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, h, context=None):
        h = self.attn1(self.norm1(h)) + h
        h = self.attn2(self.norm2(h), context=context) + h
        h = self.ff(self.norm3(h)) + h
        return h


class Transformer2DModel(nn.Module):
    """`depth` transformer blocks between a projection in and out.  SD-1.x: depth 1, 1x1-conv projections (0.8.0);
    SD-2.x / SDXL: `use_linear_projection` (nn.Linear over tokens) and, for SDXL, depth 2 / 10 [3P]."""

    def __init__(self, heads: int, dim_head: int, in_channels: int, cross_attention_dim: int, groups: int = 32,
                 depth: int = 1, use_linear_projection: bool = False):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear_projection else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)
                                                 for _ in range(depth)])
        self.proj_out = nn.Linear(inner, in_channels) if use_linear_projection else nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, context=None):
        b, c, hh, ww = x.shape
        res = x
        h = self.norm(x)
        if self.use_linear_projection:
            h = self.proj_in(h.permute(0, 2, 3, 1).reshape(b, hh * ww, c))
        else:
            h = self.proj_in(h)
            h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, h.shape[1])
        for blk in self.transformer_blocks:
            h = blk(h, context=context)
        if self.use_linear_projection:
            return self.proj_out(h).reshape(b, hh, ww, -1).permute(0, 3, 1, 2) + res
        h = h.reshape(b, hh, ww, -1).permute(0, 3, 1, 2)
        return self.proj_out(h) + res


class AttentionBlock(nn.Module):
    """Single-/multi-head spatial self-attention of the VAE mid block (0.8.0 `AttentionBlock`) [3P]."""

    def __init__(self, channels: int, num_head_channels=None, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.channels = channels
        self.num_heads = channels // num_head_channels if num_head_channels is not None else 1
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps, affine=True)
        self.query = nn.Linear(channels, channels)
        self.key = nn.Linear(channels, channels)
        self.value = nn.Linear(channels, channels)
        self.proj_attn = nn.Linear(channels, channels)

    def forward(self, x):
        b, c, hh, ww = x.shape
        h = self.group_norm(x).view(b, c, hh * ww).transpose(1, 2)
        q, k, v = self.query(h), self.key(h), self.value(h)
        nh = self.num_heads
        d = c // nh

        def split(t):
            return t.reshape(b, -1, nh, d).permute(0, 2, 1, 3).reshape(b * nh, -1, d)
        q, k, v = split(q), split(k), split(v)
        attn = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
                             q, k.transpose(1, 2), beta=0, alpha=1.0 / (d ** 0.5)).softmax(dim=-1)
        o = torch.bmm(attn, v).reshape(b, nh, -1, d).permute(0, 2, 1, 3).reshape(b, -1, c)
        o = self.proj_attn(o).transpose(1, 2).reshape(b, c, hh, ww)
        return o + x
