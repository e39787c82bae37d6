"""Arg-max helpers with the reference's names/semantics (eval.py:39-111), running on the HIP
token-statistics kernel (csrc/skp_select_loss.hip).  The rest of the reference's eval.py (dataset
metrics, plotting) is out of scope."""
from __future__ import annotations

import torch

from . import ops


def _points(flat: torch.Tensor, w: int) -> torch.Tensor:
    return torch.stack([flat // w, flat % w], dim=-1).to(torch.float32) + 0.5


def find_max_pixel(map):
    """[B,h,w] -> [B,2] = (row+0.5, col+0.5) of the first maximal element (eval.py:39-60)."""
    am, _ = ops.token_stats(map, num_subjects=1, want_kl=False)
    return _points(am[0].long(), map.shape[-1])


def find_k_max_pixels(map, num=3):
    """[B,h,w] -> [num,B,2]: repeated arg-max with 0.05*h radius masking (eval.py:62-81)."""
    am, _ = ops.token_stats(map, num_subjects=num, want_kl=False)
    return _points(am.long(), map.shape[-1])


def mask_radius(map, max_coords, radius):
    """eval.py:83-111 (elementwise helper, kept for API parity; the fused kernels mask internally)."""
    b, h, w = map.shape
    xs = torch.arange(w, device=map.device).view(1, 1, w)
    ys = torch.arange(h, device=map.device).view(1, h, 1)
    d2 = (xs - max_coords[:, 1].view(b, 1, 1)) ** 2 + (ys - max_coords[:, 0].view(b, 1, 1)) ** 2
    return map * (d2 > radius ** 2).float()


def pixel_from_weighted_avg(heatmaps, distance=5):
    """eval.py:113-155: intensity-weighted mean location within `distance` px of the arg-max (+0.5);
    like the reference this ZEROES the far pixels of `heatmaps` in place."""
    b, m, n = heatmaps.shape
    if distance != -1:
        mx = find_max_pixel(heatmaps)
        x_max, y_max = mx[:, 0].long(), mx[:, 1].long()
        x = torch.arange(0, m, device=heatmaps.device).float().view(1, m, 1)
        y = torch.arange(0, n, device=heatmaps.device).float().view(1, 1, n)
        dist = torch.sqrt((x - x_max.view(b, 1, 1)) ** 2 + (y - y_max.view(b, 1, 1)) ** 2)
        heatmaps[dist > distance] = 0.0
    total = torch.sum(heatmaps, dim=[1, 2], keepdim=True)
    norm = heatmaps / (total + 1e-6)
    x = torch.arange(0, m, device=heatmaps.device).float().view(1, m, 1)
    y = torch.arange(0, n, device=heatmaps.device).float().view(1, 1, n)
    return torch.stack([torch.sum(x * norm, dim=[1, 2]), torch.sum(y * norm, dim=[1, 2])], dim=-1) + 0.5


@torch.no_grad()
def run_image_with_context_augmented(ldm, image, context, indices, device="cuda",
                                     from_where=["down_cross", "mid_cross", "up_cross"], layers=[0, 1, 2, 3, 4, 5],
                                     augmentation_iterations=20, noise_level=-1, augment_degrees=30,
                                     augment_scale=(0.9, 1.1), augment_translate=(0.1, 0.1), visualize=False,
                                     controllers=None, num_gpus=1, save_folder="outputs", upscale_size=512,
                                     thetas=None, noise=None, shard_over_ranks=False):
    """eval.py:197-355: maps of the selected tokens averaged over random affine views of ONE image.

    Reference loop: per augmentation -> UNet forward -> collect_maps(indices, upsample_res=upscale_size) ->
    inverse-warp the maps and a ones-mask -> accumulate; result = sum/count with NaN -> 0.
    Here all views go through the network as ONE batch (early exit, fused map kernel); gather / resize / unwarp
    stay linear ops applied after the layer/head mean.  The view count is the reference's
    `(augmentation_iterations // num_gpus) * num_gpus`.

    By default every rank that calls this processes ALL views of ITS image (callers shard the images over ranks, as
    `find_best_indices` / `optimize_embedding` do): no collective.  `shard_over_ranks=True` is for the
    opposite layout -- every rank holds the SAME image: rank 0 draws all affine matrices and all noise and broadcasts
    them, rank r takes views r, r+world, ... and the un-warped sum and the coverage count ([K,S,S] each) are
    all-reduced before the division, so the ensemble is the same `n` views as on one GPU and every rank returns the
    same averaged maps; a view count that does not divide by the world size raises.
    `thetas` [n,2,3] / `noise` [n,4,h,w] inject the draws for ALL n views (parity tests)."""
    import numpy as np
    from . import dist as skp_dist
    from .invertable_transform import RandomAffineWithInverse
    if visualize:
        raise NotImplementedError("plotting is out of scope (SURVEY.md 2.1 row 14)")
    dev, controller = next(iter(controllers.items()))
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(image).permute(2, 0, 1)
    image = image.to(device=dev, dtype=torch.float32)
    n = (augmentation_iterations // num_gpus) * num_gpus
    if n < 1:
        raise ValueError(f"augmentation_iterations ({augmentation_iterations}) is smaller than num_gpus ({num_gpus})")
    world = skp_dist.world_size() if shard_over_ranks else 1
    if n % world:
        raise ValueError(f"{n} augmented views do not divide over {world} ranks")
    tr = RandomAffineWithInverse(degrees=augment_degrees, scale=augment_scale, translate=augment_translate)
    if thetas is None:
        thetas = tr.sample_theta(n)                                # host RNG, 4 uniforms per view (invertable_transform.py:42-52)
    thetas = torch.as_tensor(thetas, dtype=torch.float32)
    if noise is None:
        noise = torch.randn(n, 4, image.shape[-2] // 8, image.shape[-1] // 8, device=dev)
    noise = noise.to(dev)
    if world > 1:                                                  # one ensemble for the whole job: rank 0's draws
        thetas, noise = thetas.to(dev).contiguous(), noise.contiguous()
        skp_dist.broadcast_(thetas, 0)
        skp_dist.broadcast_(noise, 0)
        r = skp_dist.rank()
        thetas, noise = thetas[r::world].cpu(), noise[r::world]
    tot, num = _augmented_group(ldm, image[None], context, indices, controller, dev, layers, noise_level, tr, thetas, noise,
                                upscale_size, finish=(world == 1))
    if world == 1:
        return tot[0]
    return finish_augmented(tot[0], num[0][None].expand_as(tot[0]).contiguous(), reduce=True)


def _augmented_group(ldm, images, context, indices, controller, dev, layers, noise_level, tr, thetas, noise, upscale_size,
                     finish=True):
    """`images` [m,3,H,W], `thetas` [m*n,2,3] / `noise` [m*n,4,h,w] (image-major: the n views of image 0, then image 1 ...)
    -> (sum or sum/count [m,K,S,S], count [m,S,S]).  All m*n views are ONE network batch; per image one fused
    resize + un-warp + accumulate launch."""
    from . import ptp_utils
    from ._maps import collect_maps_batched
    from .invertable_transform import RandomAffineWithInverse
    m = images.shape[0]
    n = thetas.shape[0] // m
    views = tr(images.repeat_interleave(n, dim=0), theta=thetas)
    ptp_utils.find_pred_noise(ldm, views, context.to(dev), noise_level=noise_level, device=dev, noise=noise,
                              early_exit=True, controllers={dev: controller})
    idx = torch.as_tensor(indices, device=dev).long()
    maps = collect_maps_batched(controller, layers=layers, indices=idx)    # [m*n,K,R,R]: only the selected tokens' maps
    # resize R -> upscale_size, un-warp the maps and a ones mask, sum over the views: ONE kernel (the reference's four
    # [n,K,S,S] intermediates are never materialised); a single rank gets sum / count straight from it
    theta_inv = RandomAffineWithInverse.invert(thetas.to(torch.float32))
    tots, nums = [], []
    for i in range(m):
        tot, num = ops.unwarp_accumulate(maps[i * n:(i + 1) * n], theta_inv[i * n:(i + 1) * n], int(upscale_size), finish=finish)
        tots.append(tot)
        nums.append(num)
    return torch.stack(tots), torch.stack(nums)


@torch.no_grad()
def run_images_with_context_augmented(ldm, images, context, indices, controllers=None, layers=[0, 1, 2, 3, 4, 5],
                                      augmentation_iterations=20, noise_level=-1, augment_degrees=30,
                                      augment_scale=(0.9, 1.1), augment_translate=(0.1, 0.1), num_gpus=1, upscale_size=512,
                                      thetas=None, noise=None):
    """`run_image_with_context_augmented` for SEVERAL images in one network batch (the dataset loops of
    keypoint_regressor.py:160-196 / eval.py:424-446 call it once per image): images [m,3,H,W] -> [m,K,S,S].  Per image the
    result is what the single-image function returns for the same draws (`thetas` [m*n,2,3], `noise` [m*n,4,h,w],
    image-major)."""
    from .invertable_transform import RandomAffineWithInverse
    dev, controller = next(iter(controllers.items()))
    images = images.to(device=dev, dtype=torch.float32)
    m = images.shape[0]
    n = (augmentation_iterations // num_gpus) * num_gpus
    if n < 1:
        raise ValueError(f"augmentation_iterations ({augmentation_iterations}) is smaller than num_gpus ({num_gpus})")
    tr = RandomAffineWithInverse(degrees=augment_degrees, scale=augment_scale, translate=augment_translate)
    if thetas is None:
        thetas = tr.sample_theta(m * n)
    thetas = torch.as_tensor(thetas, dtype=torch.float32)
    if noise is None:
        noise = torch.randn(m * n, 4, images.shape[-2] // 8, images.shape[-1] // 8, device=dev)
    tot, _ = _augmented_group(ldm, images, context, indices, controller, dev, layers, noise_level, tr, thetas, noise.to(dev),
                              upscale_size, finish=True)
    return tot


def finish_augmented(tot, num, reduce=True):
    """eval.py:343-353; with `reduce` across ranks: SUM all-reduce of the un-warped map sum and of the coverage count
    (one exchange of 2 x [K,S,S] floats), then sum / count with 0/0 -> 0."""
    from . import dist as skp_dist
    if reduce and skp_dist.world_size() > 1:
        both = torch.stack([tot, num])
        skp_dist.allreduce_sum_(both)
        tot, num = both[0], both[1]
    out = tot / num
    out[out != out] = 0
    return out
