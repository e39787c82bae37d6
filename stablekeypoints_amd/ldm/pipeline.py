"""`StableDiffusionPipeline` stand-in: the duck type `load_ldm` hands to the rest of the path
(SURVEY.md 8(b) "ldm duck type"): `.unet`, `.vae`, `.scheduler`, `.text_encoder`.

Architectures (`ARCHS`): SD-1.x (`sd15`), SD-2.x (`sd21`: ctx 1024, 5/10/20/20 heads of 64, linear projections),
SDXL-base (`sdxl`: three blocks, 1/2/10 transformer layers, ctx 2048, text_time micro-conditioning) and reduced-width
copies of each topology for CPU tests (`tiny`, `tiny-sd21`, `tiny-sdxl`).

Weights.  There is no network, so checkpoints come from a LOCAL DIRECTORY only:
  * `<dir>/unet.pt`, `<dir>/vae.pt` (state dicts with the diffusers keys), or the diffusers layout
    `<dir>/unet/diffusion_pytorch_model.{safetensors,bin}` (+ `config.json`), `<dir>/vae/...`;
  * loaded with `weights_only=True` and checked key by key: a missing or shape-mismatched key RAISES (the only
    tolerated leftovers are the VAE decoder half, which is not on this path).
SEEDED SYNTHETIC weights (PyTorch default inits under a fixed CPU generator seed => identical on every box) are built
only when the caller names them explicitly: `synthetic-<arch>`, `<arch>` from `ARCHS`, or any `tiny*` name.  A hub id
such as `sd-legacy/stable-diffusion-v1-5` that is not a local directory raises `FileNotFoundError` -- optimising
keypoints against random weights by accident is never silent.  `SKP_ALLOW_SYNTHETIC=1` downgrades that to a warning.
"""
from __future__ import annotations

import json
import os
import warnings

import torch
import torch.nn as nn

from .scheduler import DDIMScheduler
from .unet import UNet2DConditionModel
from .vae import AutoencoderKL

_XL_BLOCKS = dict(down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                  up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"))
ARCHS = {
    "sd15": dict(unet=dict(), vae=dict()),
    "sd21": dict(unet=dict(attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True),
                 vae=dict()),
    "sdxl": dict(unet=dict(block_out_channels=(320, 640, 1280), transformer_layers_per_block=(1, 2, 10),
                           attention_head_dim=(5, 10, 20), cross_attention_dim=2048, use_linear_projection=True,
                           addition_embed_type="text_time", addition_time_embed_dim=256,
                           projection_class_embeddings_input_dim=2816, **_XL_BLOCKS), vae=dict()),
    # reduced widths, same topology (CPU tests / smoke)
    "tiny": dict(unet=dict(block_out_channels=(32, 64, 64, 64), attention_head_dim=4, cross_attention_dim=768),
                 vae=dict(block_out_channels=(32, 32, 32, 32))),
    "tiny-sd21": dict(unet=dict(block_out_channels=(32, 64, 64, 64), attention_head_dim=(2, 4, 4, 4), cross_attention_dim=96,
                                use_linear_projection=True), vae=dict(block_out_channels=(32, 32, 32, 32))),
    "tiny-sdxl": dict(unet=dict(block_out_channels=(32, 64, 64), transformer_layers_per_block=(1, 2, 4),
                                attention_head_dim=(2, 4, 4), cross_attention_dim=128, use_linear_projection=True,
                                addition_embed_type="text_time", addition_time_embed_dim=8,
                                projection_class_embeddings_input_dim=64, **_XL_BLOCKS),
                      vae=dict(block_out_channels=(32, 32, 32, 32))),
}
TINY, SD15 = ARCHS["tiny"], ARCHS["sd15"]          # older names


def synthetic_arch(name: str):
    """Architecture key if `name` explicitly asks for seeded synthetic weights, else None."""
    n = str(name)
    if n.startswith("synthetic-"):
        n = n[len("synthetic-"):]
    if n in ARCHS:
        return n
    if n.startswith("tiny"):
        return "tiny"
    return None


def guess_arch(name: str) -> str:
    """Architecture of a hub id / directory name (used for checkpoints without a config.json)."""
    n = str(name).lower()
    if "xl" in n:
        return "sdxl"
    if "stable-diffusion-2" in n or "sd2" in n or "sd-2" in n:
        return "sd21"
    return "sd15"


class _NoTextEncoder(nn.Module):
    """The learned embedding replaces CLIP on this path; kept so `ldm.text_encoder.parameters()` works
    (optimize_token.py:73-74)."""

    def forward(self, *a, **k):
        raise RuntimeError("the text encoder is not on the token-optimisation path")


def _read_state(path_base: str):
    """state dict from `<path_base>.pt` or the diffusers files under `<path_base>/`; None when absent."""
    if os.path.exists(path_base + ".pt"):
        return torch.load(path_base + ".pt", map_location="cpu", weights_only=True)
    st = os.path.join(path_base, "diffusion_pytorch_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st, device="cpu")
    bn = os.path.join(path_base, "diffusion_pytorch_model.bin")
    if os.path.exists(bn):
        return torch.load(bn, map_location="cpu", weights_only=True)
    return None


def _unet_kwargs_from_config(path: str):
    """diffusers `unet/config.json` -> UNet2DConditionModel kwargs (only the keys this tree understands)."""
    with open(path) as f:
        c = json.load(f)
    kw = {}
    for k in ("in_channels", "out_channels", "block_out_channels", "layers_per_block", "attention_head_dim",
              "cross_attention_dim", "down_block_types", "up_block_types", "transformer_layers_per_block",
              "use_linear_projection", "addition_embed_type", "addition_time_embed_dim",
              "projection_class_embeddings_input_dim"):
        if c.get(k) is not None:
            kw[k] = tuple(c[k]) if isinstance(c[k], list) else c[k]
    return kw


def load_checked(module: nn.Module, state, what: str, allowed_unexpected=()):
    """load_state_dict with a full report: missing / mismatched keys raise; unexpected keys raise unless their prefix is
    in `allowed_unexpected` (the VAE decoder half)."""
    own = module.state_dict()
    missing = [k for k in own if k not in state]
    bad = [k for k in own if k in state and tuple(state[k].shape) != tuple(own[k].shape)]
    extra = [k for k in state if k not in own and not any(k.startswith(p) for p in allowed_unexpected)]
    if missing or bad or extra:
        raise RuntimeError(f"{what}: checkpoint does not match the module tree -- {len(missing)} missing "
                           f"(e.g. {missing[:3]}), {len(bad)} shape mismatches (e.g. {bad[:3]}), {len(extra)} unexpected "
                           f"(e.g. {extra[:3]})")
    module.load_state_dict({k: state[k] for k in own}, strict=True)


class StableDiffusionPipeline:
    def __init__(self, unet, vae, scheduler, arch="sd15", synthetic=False):
        self.unet, self.vae, self.scheduler = unet, vae, scheduler
        self.text_encoder = _NoTextEncoder()
        self.device = torch.device("cpu")
        self.arch, self.synthetic_weights = arch, synthetic

    @staticmethod
    def build(arch: str, seed=0, unet_kwargs=None, init_device=None):
        """Module trees with seeded default inits.  `init_device=None`: parameters are drawn on the CPU (identical on
        every box, and on CPU and GPU runs -- what the parity tests rely on).  `init_device="cuda:N"`: drawn directly
        on that GPU from its own seeded generator (identical on every rank of a job, different numbers from the CPU
        draw): N ranks of a multi-GPU run do not contend for the host cores with N x 3.4 GB of CPU-side init."""
        cfg = ARCHS[arch]
        on_gpu = init_device is not None and torch.device(init_device).type == "cuda"
        gen_state = torch.random.get_rng_state()
        cuda_state = torch.cuda.get_rng_state(init_device) if on_gpu else None
        torch.manual_seed(seed)
        try:
            if on_gpu:
                with torch.device(init_device):
                    unet = UNet2DConditionModel(**(unet_kwargs if unet_kwargs is not None else cfg["unet"]))
                    vae = AutoencoderKL(**cfg["vae"])
            else:
                unet = UNet2DConditionModel(**(unet_kwargs if unet_kwargs is not None else cfg["unet"]))
                vae = AutoencoderKL(**cfg["vae"])
        finally:
            torch.random.set_rng_state(gen_state)
            if on_gpu:
                torch.cuda.set_rng_state(cuda_state, init_device)
        return unet, vae

    @classmethod
    def from_pretrained(cls, type="sd-legacy/stable-diffusion-v1-5", use_auth_token=None, scheduler=None, seed=0,
                        init_device=None):
        name = str(type)
        sched = scheduler if scheduler is not None else DDIMScheduler()
        if os.path.isdir(name):
            cfg_path = os.path.join(name, "unet", "config.json")
            arch = guess_arch(os.path.basename(os.path.normpath(name)))
            kw = _unet_kwargs_from_config(cfg_path) if os.path.exists(cfg_path) else None
            unet, vae = cls.build(arch, seed, kw)
            for sub, mod, allowed in (("unet", unet, ()), ("vae", vae, ("decoder.", "post_quant_conv."))):
                state = _read_state(os.path.join(name, sub))
                if state is None:
                    raise FileNotFoundError(f"{name}: no {sub}.pt and no {sub}/diffusion_pytorch_model.(safetensors|bin)")
                load_checked(mod, state, f"{name}/{sub}", allowed)
            unet.eval(); vae.eval()
            return cls(unet, vae, sched, arch, synthetic=False)
        arch = synthetic_arch(name)
        if arch is None:
            if os.environ.get("SKP_ALLOW_SYNTHETIC") != "1":
                raise FileNotFoundError(
                    f"'{name}' is not a local checkpoint directory (this build has no network access to the hub). "
                    f"Pass a directory with unet/vae weights, or ask for seeded synthetic weights explicitly with "
                    f"'synthetic-{guess_arch(name)}' (or SKP_ALLOW_SYNTHETIC=1).")
            arch = guess_arch(name)
            warnings.warn(f"'{name}' not found locally: building the {arch} architecture with SEEDED SYNTHETIC weights "
                          "(SKP_ALLOW_SYNTHETIC=1) -- keypoints optimised against it are meaningless", stacklevel=2)
        unet, vae = cls.build(arch, seed, init_device=init_device)
        unet.eval(); vae.eval()
        return cls(unet, vae, sched, arch, synthetic=True)

    def to(self, device):
        self.device = torch.device(device)
        self.unet.to(self.device)
        self.vae.to(self.device)
        return self
