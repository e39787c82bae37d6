"""The reference's optimisation step on CPU, end to end (TEST INFRASTRUCTURE + bench cpu_baseline).

Runs the build's SD-shaped UNet/VAE module tree (the stand-in for diffusers, which is absent
offline) on the host with the REFERENCE's hooked attention restated in oracle/ref_path.py:
materialised softmax + clone, bicubic up-sampling of the layer input, second to_q, per-head
(B*h, R^2, T) stores, stack+mean `collect_maps`, python-loop selection, torch losses -- i.e. the
op sequence of ptp_utils.py:480-541 / optimize.py:27-79,339-425 with the only deviation the
reference itself needs on a CPU (`.to('cuda:0')`, optimize.py:405-406).
"""
from __future__ import annotations

import time

import torch

from . import ref_path as R


def register_reference_hook(unet, store: R.OracleStore, feature_upsample_res: int, max_seq: int = 32 ** 2) -> int:
    """ptp_utils.py:555-573 restated: patch class-named `CrossAttention` modules under '*up*' children.
    `max_seq` is the reference's hard-coded 32**2 gate (ptp_utils.py:510); tests shrink it to reproduce, on a small
    latent grid, the SD-2.x @768^2 situation where only three layers pass the gate."""
    count = 0

    def patch(mod):
        to_out = mod.to_out[0] if isinstance(mod.to_out, torch.nn.ModuleList) else mod.to_out

        def forward(x, context=None, mask=None):
            return R.hooked_attention(x, context, mod.to_q.weight, mod.to_k.weight, mod.to_v.weight,
                                      to_out.weight, to_out.bias, mod.heads, store, feature_upsample_res,
                                      max_seq=max_seq)
        mod.forward = forward

    def rec(net):
        nonlocal count
        if net.__class__.__name__ == "CrossAttention":
            patch(net)
            count += 1
        else:
            for c in net.children():
                rec(c)

    for name, child in unet.named_children():
        if "up" in name:
            rec(child)
    store.num_att_layers = count
    assert count != 0
    return count


def maps_for(ldm, image, context, store, noise, layers=(0, 1, 2, 3), noise_level=-1):
    """ptp_utils.py:205-272 for one image [1,3,H,W]: latent -> add_noise -> full UNet forward -> collect_maps."""
    with torch.no_grad():
        latent = ldm.vae.encode(image * 2 - 1)["latent_dist"].mean * 0.18215
    t = ldm.scheduler.timesteps[noise_level]
    noisy = ldm.scheduler.add_noise(latent, noise, t)
    ldm.unet(noisy, t.repeat(noisy.shape[0]), context.repeat(noisy.shape[0], 1, 1))
    return R.collect_maps(store, upsample_res=-1, layers=layers)


def image_step(ldm, image, context, store, theta, noise, noise_t, *, layers=(0, 1, 2, 3), sigma=2.0,
               furthest_point_num_samples=25, top_k=10, num_subjects=1, w_sharp=100.0, w_equiv=1000.0,
               top_k_strategy="gaussian"):
    """optimize.py:347-414 for one image (G=1): returns (loss, sharp, equiv, sel, map, map_t)."""
    am = maps_for(ldm, image, context, store, noise, layers)
    warped = R.affine_warp(image, theta)
    am_t = maps_for(ldm, warped, context, store, noise_t, layers)
    loss, sharp, equiv, sel = R.image_loss(am, am_t, theta, 0, furthest_point_num_samples=furthest_point_num_samples,
                                           top_k=top_k, sigma=sigma, num_subjects=num_subjects,
                                           sharpening_loss_weight=w_sharp, equivariance_attn_loss_weight=w_equiv,
                                           top_k_strategy=top_k_strategy)
    return loss, sharp, equiv, sel, am, am_t


def optimize_embedding_cpu(ldm, images, context, *, steps, batch_size, R_up, lr=5e-3, seed=0, **kw):
    """optimize.py:269-452 on CPU for `steps` optimizer steps of `batch_size` images each (num_gpus=1).
    Returns (context, seconds, images_processed); timing covers the loop only."""
    store = R.OracleStore()
    register_reference_hook(ldm.unet, store, R_up)
    context = context.clone().requires_grad_(True)
    opt = torch.optim.Adam([context], lr=lr)
    g = torch.Generator().manual_seed(seed)
    n_img = 0
    t0 = time.perf_counter()
    for step in range(steps):
        for j in range(batch_size):
            img = images[(step * batch_size + j) % images.shape[0]][None]
            a, sc, tr = R.draw_affine_params(lambda: torch.rand(1, generator=g).item(), 15, (0.8, 1.0), (0.25, 0.25))
            theta = R.affine_matrix(a, sc, tr)
            lat_shape = (1, 4, img.shape[-2] // 8, img.shape[-1] // 8)
            noise = torch.randn(lat_shape, generator=g)
            noise_t = torch.randn(lat_shape, generator=g)
            loss, *_ = image_step(ldm, img, context, store, theta, noise, noise_t, **kw)
            (loss / batch_size).backward()
            n_img += 1
        opt.step()
        opt.zero_grad()
    return context.detach(), time.perf_counter() - t0, n_img


def optimize_trajectory(ldm, images, context, order, noise, thetas, *, steps, accum, R_up, lr=5e-3, **kw):
    """optimize.py:339-425 with the loader order and every random draw INJECTED (G11): `order` [steps*accum] image
    indices, `noise` [2*steps*accum, 4,h,w] in draw order (image view, warped view, ...), `thetas` [steps*accum,2,3].
    Returns the embedding after every optimizer step, [steps, T, C]."""
    store = R.OracleStore()
    register_reference_hook(ldm.unet, store, R_up)
    context = context.clone().requires_grad_(True)
    opt = torch.optim.Adam([context], lr=lr)
    after = []
    for it in range(steps * accum):
        img = images[int(order[it])][None]
        loss, *_ = image_step(ldm, img, context, store, thetas[it:it + 1], noise[2 * it:2 * it + 1],
                              noise[2 * it + 1:2 * it + 2], **kw)
        (loss / accum).backward()                               # optimize.py:418-420
        if (it + 1) % accum == 0:                               # :421-423
            opt.step()
            opt.zero_grad()
            after.append(context.detach().clone()[0])
    return torch.stack(after)


def find_best_indices(ldm, images, context, order, noise, *, R_up, furthest_point_num_samples, top_k, sigma,
                      num_subjects=1, layers=(0, 1, 2, 3), with_scores=False):
    """keypoint_regressor.py:56-108 with the loader order and the noise draws injected (G12): per image ONE untransformed
    view -> candidates by KL -> furthest-point sampling on the SAME map -> vote.  -> (indices, per-image selections)."""
    store = R.OracleStore()
    register_reference_hook(ldm.unet, store, R_up)
    picked, scores = [], []
    with torch.no_grad():
        for it, i in enumerate(order):
            am = maps_for(ldm, images[int(i)][None], context, store, noise[it:it + 1], layers)
            cand = R.find_top_k_gaussian(am, furthest_point_num_samples, sigma=sigma, num_subjects=num_subjects)
            picked.append(R.furthest_point_sampling(am, top_k, cand))
            if with_scores:
                scores.append(R.gaussian_kl(am, sigma, num_subjects=num_subjects))
    flat = torch.cat(picked)                                                                     # :100-101
    indices, counts = torch.unique(flat, return_counts=True)                                     # :103
    out = indices[counts.argsort(descending=True)][:top_k]                                       # :104-105
    return (out, torch.stack(picked), torch.stack(scores)) if with_scores else (out, torch.stack(picked))
