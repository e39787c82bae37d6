"""`StableDiffusionPipeline` stand-in: the duck type `load_ldm` hands to the rest of the path
(SURVEY.md 8(b) "ldm duck type"): `.unet`, `.vae`, `.scheduler`, `.text_encoder`.

Weights: there is no network, so `from_pretrained` builds the SD-1.x architecture with SEEDED
synthetic weights (PyTorch default inits under a fixed CPU generator seed => identical on every
box and on CPU/GPU).  If `type` is a local directory holding `unet.pt` / `vae.pt` state dicts with
the diffusers 0.8.0 keys they are loaded instead.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .scheduler import DDIMScheduler
from .unet import UNet2DConditionModel
from .vae import AutoencoderKL

# small configs for CPU tests (same topology, reduced widths)
TINY = dict(unet=dict(block_out_channels=(32, 64, 64, 64), attention_head_dim=4, cross_attention_dim=768),
            vae=dict(block_out_channels=(32, 32, 32, 32)))
SD15 = dict(unet=dict(), vae=dict())


class _NoTextEncoder(nn.Module):
    """The learned embedding replaces CLIP on this path; kept so `ldm.text_encoder.parameters()` works
    (optimize_token.py:73-74)."""

    def forward(self, *a, **k):
        raise RuntimeError("the text encoder is not on the token-optimisation path")


class StableDiffusionPipeline:
    def __init__(self, unet, vae, scheduler):
        self.unet, self.vae, self.scheduler = unet, vae, scheduler
        self.text_encoder = _NoTextEncoder()
        self.device = torch.device("cpu")

    @classmethod
    def from_pretrained(cls, type="sd-legacy/stable-diffusion-v1-5", use_auth_token=None, scheduler=None, seed=0):
        cfg = TINY if str(type).startswith("tiny") else SD15
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        try:
            unet = UNet2DConditionModel(**cfg["unet"])
            vae = AutoencoderKL(**cfg["vae"])
        finally:
            torch.random.set_rng_state(gen_state)
        if os.path.isdir(str(type)):
            for name, mod in (("unet", unet), ("vae", vae)):
                p = os.path.join(str(type), name + ".pt")
                if os.path.exists(p):
                    mod.load_state_dict(torch.load(p, map_location="cpu"), strict=False)
        unet.eval(); vae.eval()
        return cls(unet, vae, scheduler if scheduler is not None else DDIMScheduler())

    def to(self, device):
        self.device = torch.device(device)
        self.unet.to(self.device)
        self.vae.to(self.device)
        return self
