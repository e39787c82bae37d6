"""Build-owned SD-1.x `UNet2DConditionModel` with the diffusers==0.8.0 module tree [3P].

Why it exists: the reference's hot path lives INSIDE diffusers' UNet (ptp_utils.py:227-229), and
diffusers / the SD weights are not available offline.  This module reproduces the architecture of
the published `unet/config.json` of Stable Diffusion 1.x (block_out_channels 320/640/1280/1280,
2 layers per block, 8 heads, cross_attention_dim 768, GroupNorm-32) with the 0.8.0 state-dict
keys (`up_blocks.1.attentions.0.transformer_blocks.0.attn2.to_q.weight`, ...) so that
(a) the reference's name-based patcher finds exactly the same modules and (b) a real checkpoint
loads by key.  Values are "parity unpinned" (DESIGN.md section 3); structure is tested.

`forward(sample, timestep, encoder_hidden_states)` returns {"sample": eps} like 0.8.0's
`UNet2DConditionOutput` indexed with ["sample"] (ptp_utils.py:229).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import Transformer2DModel


class StopForward(Exception):
    """Raised by the hooked attention once every requested map is stored (early exit)."""


def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, out_dim)
        self.linear_2 = nn.Linear(out_dim, out_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, temb_ch=1280, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_ch, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, out_ch) if temb_ch is not None else None
        self.norm2 = nn.GroupNorm(groups, out_ch, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_ch, out_ch, 1) if in_ch != out_ch else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None and temb is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, ch: int, padding: int = 1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, in_ch, out_ch, temb_ch, heads, ctx_dim, num_layers=2, add_downsample=True, depth=1, linear_proj=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, temb_ch) for i in range(num_layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, out_ch // heads, out_ch, ctx_dim, depth=depth,
                                                            use_linear_projection=linear_proj) for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_downsample else None

    def forward(self, h, temb, ctx):
        outs = ()
        for r, a in zip(self.resnets, self.attentions):
            h = a(r(h, temb), context=ctx)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class DownBlock2D(nn.Module):
    def __init__(self, in_ch, out_ch, temb_ch, num_layers=2, add_downsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, temb_ch) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_downsample else None

    def forward(self, h, temb, ctx=None):
        outs = ()
        for r in self.resnets:
            h = r(h, temb)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, ch, temb_ch, heads, ctx_dim, depth=1, linear_proj=False):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(heads, ch // heads, ch, ctx_dim, depth=depth,
                                                            use_linear_projection=linear_proj)])
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch), ResnetBlock2D(ch, ch, temb_ch)])

    def forward(self, h, temb, ctx):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, context=ctx)
        return self.resnets[1](h, temb)


class UpBlock2D(nn.Module):
    def __init__(self, in_ch, prev_ch, out_ch, temb_ch, num_layers=3, add_upsample=True):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_ch if i == num_layers - 1 else out_ch
            res.append(ResnetBlock2D((prev_ch if i == 0 else out_ch) + skip, out_ch, temb_ch))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None

    def forward(self, h, skips, temb, ctx=None):
        for r in self.resnets:
            h = r(torch.cat([h, skips.pop()], dim=1), temb)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, in_ch, prev_ch, out_ch, temb_ch, heads, ctx_dim, num_layers=3, add_upsample=True, depth=1,
                 linear_proj=False):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_ch if i == num_layers - 1 else out_ch
            res.append(ResnetBlock2D((prev_ch if i == 0 else out_ch) + skip, out_ch, temb_ch))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList([Transformer2DModel(heads, out_ch // heads, out_ch, ctx_dim, depth=depth,
                                                            use_linear_projection=linear_proj) for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None

    def forward(self, h, skips, temb, ctx):
        for r, a in zip(self.resnets, self.attentions):
            h = a(r(torch.cat([h, skips.pop()], dim=1), temb), context=ctx)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


def _per_block(v, n):
    return [int(v)] * n if isinstance(v, int) else [int(x) for x in v]


class UNet2DConditionModel(nn.Module):
    """SD-1.x by default.  SD-2.x: `attention_head_dim=(5,10,20,20)` (head COUNT per block, 64-wide heads),
    `cross_attention_dim=1024`, `use_linear_projection=True`.  SDXL-base: three blocks (320,640,1280) with
    `transformer_layers_per_block=(1,2,10)`, heads (5,10,20), `cross_attention_dim=2048` and the
    `addition_embed_type="text_time"` micro-conditioning branch (pooled text embedding + 6 size/crop ids) [3P]."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels: Sequence[int] = (320, 640, 1280, 1280),
                 layers_per_block=2, attention_head_dim=8, cross_attention_dim=768,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 transformer_layers_per_block=1, use_linear_projection=False, addition_embed_type=None,
                 addition_time_embed_dim=None, projection_class_embeddings_input_dim=None):
        super().__init__()
        boc = list(block_out_channels)
        nb = len(boc)
        temb = boc[0] * 4
        heads = _per_block(attention_head_dim, nb)     # 0.8.0: `attention_head_dim` is the head COUNT [3P]
        depth = _per_block(transformer_layers_per_block, nb)
        lin = bool(use_linear_projection)
        self.config = dict(in_channels=in_channels, block_out_channels=tuple(boc), cross_attention_dim=cross_attention_dim,
                           addition_embed_type=addition_embed_type)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self._t_dim = boc[0]
        self.add_embedding = None
        if addition_embed_type == "text_time":
            self._add_t_dim = int(addition_time_embed_dim)
            self.add_embedding = TimestepEmbedding(int(projection_class_embeddings_input_dim), temb)
        elif addition_embed_type is not None:
            raise NotImplementedError(addition_embed_type)
        downs, ch = [], boc[0]
        for i, kind in enumerate(down_block_types):
            last = i == nb - 1
            if kind == "CrossAttnDownBlock2D":
                downs.append(CrossAttnDownBlock2D(ch, boc[i], temb, heads[i], cross_attention_dim, layers_per_block, not last,
                                                  depth=depth[i], linear_proj=lin))
            else:
                downs.append(DownBlock2D(ch, boc[i], temb, layers_per_block, not last))
            ch = boc[i]
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], temb, heads[-1], cross_attention_dim, depth=depth[-1], linear_proj=lin)
        ups, rev, rheads, rdepth = [], boc[::-1], heads[::-1], depth[::-1]
        prev = rev[0]
        for i, kind in enumerate(up_block_types):
            out_ch = rev[i]
            in_ch = rev[min(i + 1, nb - 1)]
            last = i == nb - 1
            if kind == "CrossAttnUpBlock2D":
                ups.append(CrossAttnUpBlock2D(in_ch, prev, out_ch, temb, rheads[i], cross_attention_dim, layers_per_block + 1,
                                              not last, depth=rdepth[i], linear_proj=lin))
            else:
                ups.append(UpBlock2D(in_ch, prev, out_ch, temb, layers_per_block + 1, not last))
            prev = out_ch
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def default_added_cond(self, sample):
        """SDXL micro-conditioning when the caller gives none: zero pooled text embedding (no text encoder on this path)
        and time_ids = (orig_h, orig_w, crop_top, crop_left, target_h, target_w) of the un-cropped input image."""
        b = sample.shape[0]
        hh, ww = float(sample.shape[-2] * 8), float(sample.shape[-1] * 8)
        pooled = self.add_embedding.linear_1.in_features - 6 * self._add_t_dim
        return {"text_embeds": torch.zeros(b, pooled, device=sample.device, dtype=sample.dtype),
                "time_ids": torch.tensor([[hh, ww, 0.0, 0.0, hh, ww]], device=sample.device, dtype=sample.dtype).expand(b, -1)}

    def time_path(self, sample, timestep, added_cond_kwargs=None):
        """Time (+ SDXL micro-conditioning) embedding [rows, temb_ch]: independent of the latents and of the text embedding."""
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=sample.device)
        timestep = timestep.reshape(-1).to(sample.device).expand(sample.shape[0])
        temb = self.time_embedding(timestep_embedding(timestep, self._t_dim).to(sample.dtype))
        if self.add_embedding is not None:
            cond = added_cond_kwargs if added_cond_kwargs is not None else self.default_added_cond(sample)
            ids = cond["time_ids"]
            tids = timestep_embedding(ids.reshape(-1), self._add_t_dim).reshape(ids.shape[0], -1)
            temb = temb + self.add_embedding(torch.cat([cond["text_embeds"], tids.to(sample.dtype)], dim=-1))
        return temb

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None):
        temb = self.time_path(sample, timestep, added_cond_kwargs)
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, temb, encoder_hidden_states)
            skips += list(outs)
        h = self.mid_block(h, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            h = blk(h, skips, temb, encoder_hidden_states)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return {"sample": h}
