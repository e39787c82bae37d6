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
def find_best_indices(ldm, context, args, controllers, num_gpus, from_where=["down_cross", "mid_cross", "up_cross"]):
    """Same selection per image as the training loop but on ONE (untransformed) view
    (keypoint_regressor.py:70-106); images are sharded over ranks, the vote is global."""
    world, rank = skp_dist.world_size(), skp_dist.rank()
    width = num_gpus * world
    dev, controller = next(iter(controllers.items()))
    dataset = build_dataset(args)
    n_iter = args.num_indices // width
    gen = torch.Generator().manual_seed(getattr(args, "seed", 0) + 4321)
    order = skp_dist.shard_indices(torch.randperm(len(dataset), generator=gen).tolist(), rank, world)
    group = max(1, getattr(args, "images_per_forward", 8))
    picked, done = [], 0
    while done < n_iter:
        n = min(group, n_iter - done)
        imgs = torch.stack([dataset[order[(done + i) % len(order)]]["img"] for i in range(n)]).to(dev)
        ptp_utils.find_pred_noise(ldm, imgs, context.to(dev), noise_level=args.noise_level, device=dev,
                                  early_exit=True, controllers={dev: controller})
        maps = collect_maps_batched(controller, layers=args.layers)
        for i in range(n):
            am, order = token_order(maps[i], getattr(args, "top_k_strategy", "gaussian"),
                                    getattr(args, "num_subjects", 1), args.sigma)
            n_cand = min(args.furthest_point_num_samples, maps.shape[1])
            _, sel = ops.select_tokens(order, am[0], maps.shape[-1], n_cand, args.top_k)
            picked.append(sel)
        done += n
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
