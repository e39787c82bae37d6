"""Build-owned SD `AutoencoderKL` ENCODER half with the diffusers==0.8.0 module tree [3P].

Only `encode(x)["latent_dist"].mean` is on the reference's hot path (ptp_utils.py:289-304); the
decoder belongs to the out-of-scope image-generation demo and is not built.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import AttentionBlock
from .unet import ResnetBlock2D, Downsample2D


class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_ch, out_ch, num_layers=2, add_downsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, temb_ch=None, eps=1e-6)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch, padding=0)]) if add_downsample else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
        return h


class UNetMidBlock2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.attentions = nn.ModuleList([AttentionBlock(ch, None, groups=32, eps=1e-6)])
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch=None, eps=1e-6), ResnetBlock2D(ch, ch, temb_ch=None, eps=1e-6)])

    def forward(self, h):
        h = self.resnets[0](h)
        h = self.attentions[0](h)
        return self.resnets[1](h)


class Encoder(nn.Module):
    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        boc = list(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        blocks, ch = [], boc[0]
        for i, oc in enumerate(boc):
            blocks.append(DownEncoderBlock2D(ch, oc, layers_per_block, add_downsample=i != len(boc) - 1))
            ch = oc
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = UNetMidBlock2D(boc[-1])
        self.conv_norm_out = nn.GroupNorm(32, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for b in self.down_blocks:
            h = b(h)
        h = self.mid_block(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class DiagonalGaussianDistribution:
    def __init__(self, parameters: torch.Tensor):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)


class AutoencoderKL(nn.Module):
    def __init__(self, latent_channels=4, block_out_channels=(128, 256, 512, 512)):
        super().__init__()
        self.encoder = Encoder(3, latent_channels, block_out_channels)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    def encode(self, x):
        return {"latent_dist": DiagonalGaussianDistribution(self.quant_conv(self.encoder(x)))}
