"""`find_best_indices` (keypoint_regressor.py:16-108): vote the most frequently selected tokens over
`num_indices` images.  The regressors / dataset-bound `precompute_all_keypoints` are out of scope
(SURVEY.md 2.1 row 12); `keypoints_from_maps` is the map -> location step they use (:191-196)."""
from __future__ import annotations

import torch

from . import dist as skp_dist
from . import ops, ptp_utils
from ._maps import collect_maps_batched
from .eval import find_max_pixel, pixel_from_weighted_avg
from .optimize import build_dataset, token_order


@torch.no_grad()
def find_best_indices(ldm, context, args, controllers, num_gpus, from_where=["down_cross", "mid_cross", "up_cross"],
                      draws=None, votes_out=None):
    """Same selection per image as the training loop but on ONE (untransformed) view
    (keypoint_regressor.py:56-98), then the `top_k` most frequently selected tokens (:100-106).  Images are sharded over
    the ranks (`num_indices // (num_gpus * world)` each, as the reference's `num_indices // num_gpus` loop), the vote
    is global.  `draws = (order, noise)` injects THIS rank's image order and the noise of every forward [n,4,h,w]
    (parity tests; the reference takes both from the global RNGs); `votes_out`, a list, receives the per-image
    selections [n_images, top_k] before the vote."""
    world, rank = skp_dist.world_size(), skp_dist.rank()
    width = num_gpus * world
    dev, controller = next(iter(controllers.items()))
    dataset = build_dataset(args)
    n_iter = args.num_indices // width
    if n_iter < 1:
        raise ValueError(f"num_indices ({args.num_indices}) is smaller than the data-parallel width ({width})")
    if draws is not None:
        image_order, noise = [int(i) for i in draws[0]], draws[1]
    else:
        gen = torch.Generator().manual_seed(getattr(args, "seed", 0) + 4321)
        image_order, noise = [], None
        while len(image_order) < n_iter:                          # reshuffled every epoch, like the reference's loader
            image_order += skp_dist.shard_indices(torch.randperm(len(dataset), generator=gen).tolist(), rank, world)
    if len(image_order) < n_iter:
        raise ValueError("find_best_indices: fewer images than num_indices // width")
    strategy, ns = getattr(args, "top_k_strategy", "gaussian"), getattr(args, "num_subjects", 1)
    group = max(1, getattr(args, "images_per_forward", 8))
    picked, done = [], 0
    while done < n_iter:
        n = min(group, n_iter - done)
        imgs = torch.stack([dataset[image_order[done + i]]["img"] for i in range(n)]).to(dev)
        ptp_utils.find_pred_noise(ldm, imgs, context.to(dev), noise_level=args.noise_level, device=dev,
                                  noise=None if noise is None else noise[done:done + n].to(dev),
                                  early_exit=True, controllers={dev: controller})
        maps = collect_maps_batched(controller, layers=args.layers)
        # the greedy loop returns min(top_k, candidates) tokens (ptp_utils.py:142-157)
        n_cand = min(args.furthest_point_num_samples, maps.shape[1])
        for i in range(n):
            am, score = token_order(maps[i], strategy, ns, args.sigma)
            _, sel = ops.select_tokens(score, am[0], maps.shape[-1], n_cand, min(args.top_k, n_cand))
            picked.append(sel)
        done += n
    if votes_out is not None:
        votes_out.append(torch.stack(picked).cpu())
    picked = torch.cat(picked)
    if world > 1:
        allp = [torch.empty_like(picked) for _ in range(world)]
        torch.distributed.all_gather(allp, picked)
        picked = torch.cat(allp)
    indices, counts = torch.unique(picked.cpu(), return_counts=True)
    return indices[counts.argsort(descending=True)][:args.top_k]


def keypoints_from_maps(attention_maps, max_loc_strategy="argmax"):
    """keypoint_regressor.py:191-196: [K,S,S] -> [K,2] (row, col) in [0,1]."""
    size = float(attention_maps.shape[-1])
    if max_loc_strategy == "argmax":
        return find_max_pixel(attention_maps) / size
    return pixel_from_weighted_avg(attention_maps) / size
