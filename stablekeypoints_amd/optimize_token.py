"""`load_ldm` and the gaussian targets, reference API of optimize_token.py:24-78,203-241.

MI355X design: ONE PROCESS PER GPU (torch.distributed over RCCL) instead of `nn.DataParallel`
(optimize_token.py:41-50), so there is no per-forward parameter re-broadcast and no re-patching
hook: the frozen UNet/VAE live on this rank's GPU for the whole run, `controllers` has the single
entry {this rank's device: AttentionStore()} and `effective_num_gpus` is the number of devices THIS
PROCESS drives (1).  The data-parallel width comes from `torch.distributed.get_world_size()`.
"""
from __future__ import annotations

import torch

from . import ptp_utils
from .ldm.pipeline import StableDiffusionPipeline
from .ldm.scheduler import DDIMScheduler


def load_ldm(device, type="CompVis/stable-diffusion-v1-4", feature_upsample_res=256, my_token=None,
             init_on_device=False):
    """optimize_token.py:24-78.  `init_on_device` (extension, synthetic weights only): draw the seeded weights directly
    on `device` instead of on the host (multi-GPU start-up; see StableDiffusionPipeline.build)."""
    scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              clip_sample=False, set_alpha_to_one=False)
    scheduler.set_timesteps(50)                                   # NUM_DDIM_STEPS, optimize_token.py:33-34
    ldm = StableDiffusionPipeline.from_pretrained(type, use_auth_token=my_token, scheduler=scheduler,
                                                  init_device=device if init_on_device else None).to(device)
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    controllers = {dev: ptp_utils.AttentionStore()}
    # patched once: the module tree is never re-replicated (cf. the forward-pre-hook of optimize_token.py:60-69)
    ptp_utils.register_attention_control(ldm.unet, controllers[dev], feature_upsample_res=feature_upsample_res)
    ptp_utils.accelerate_cross_attention(ldm.unet)     # down/mid cross layers: same fused core, never stored
    if dev.type == "cuda":
        # the attention cores have HIP kernels for fixed head sizes and NO eager route on the GPU (ptp_utils._attention_core):
        # refuse an unsupported UNet configuration here, not in the middle of the first forward
        from . import ops
        bad = sorted({m.to_q.weight.shape[0] // m.heads for m in ldm.unet.modules() if m.__class__.__name__ == "CrossAttention"
                      and not ops.self_attn_supported(m.to_q.weight.shape[0], m.heads)})
        if bad:
            raise RuntimeError(f"load_ldm: attention head sizes {bad} have no HIP kernel (built: {ops.CROSS_ATTN_HEAD_DIMS}); "
                               "see INTEGRATION.md, 'Supported UNet configurations'")
        # GroupNorm(+bias/temb offset)+SiLU of the frozen blocks, fused
        from .ldm.fused import fuse_norms
        fuse_norms(ldm.unet)
        fuse_norms(ldm.vae)
        from . import tuning
        tuning.enable()                                 # measured GEMM algorithm choices for the frozen Linear layers
    for module in (ldm.vae, ldm.text_encoder, ldm.unet):
        for p in module.parameters():
            p.requires_grad = False
    return ldm, controllers, 1


def gaussian_circle(pos, size=64, sigma=16, device="cuda"):
    """optimize_token.py:203-225: pos [B,2] in [0,1] (row, col) -> [B,size,size]."""
    p = (pos * size).reshape(-1, 1, 1, 2)
    ar = torch.arange(size, device=pos.device)
    rows = ar.view(1, size, 1) + 0.5
    cols = ar.view(1, 1, size) + 0.5
    d2 = (cols - p[..., 1]) ** 2 + (rows - p[..., 0]) ** 2
    return torch.exp(-1 * d2 / (2.0 * sigma ** 2.0))


def gaussian_circles(pos, size=64, sigma=16, device="cuda"):
    """optimize_token.py:227-241: pos [num_points,B,2] -> mean over points."""
    return torch.stack([gaussian_circle(pos[i], size=size, sigma=sigma) for i in range(pos.shape[0])]).mean(dim=0)
