"""`find_best_indices` (keypoint_regressor.py:16-108): vote the most frequently selected tokens over
`num_indices` images, and `precompute_all_keypoints` (:111-198): the dataset-level keypoint driver -- per image the
augmented inference -> map -> location.  The regressors (`return_regressor*`, numpy least squares) are out of scope
(SURVEY.md 2.1 row 12); `keypoints_from_maps` is the map -> location step (:191-196)."""
from __future__ import annotations

import torch

from . import dist as skp_dist
from . import ops, ptp_utils
from ._maps import collect_maps_batched
from .eval import find_max_pixel, pixel_from_weighted_avg, run_images_with_context_augmented
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


@torch.no_grad()
def precompute_all_keypoints(ldm, context, top_indices, args, controllers, num_gpus,
                             from_where=["down_cross", "mid_cross", "up_cross"], dataset=None, draws=None):
    """keypoint_regressor.py:111-198 -> (source_keypoints [N,K,2], target_keypoints [N,...] or None, visibility or None)
    for the first `min(len(dataset), args.max_num_points)` images of a shuffled pass (:155-165).

    Reference loop: per image `run_image_with_context_augmented` (`augmentation_iterations` affine views, one network
    forward each) -> 512 x 512 maps of the voted tokens (the default `upscale_size` of eval.py:213, hard-wired again in
    the `/ 512.0` of :192-195) -> arg-max or intensity-weighted location.  Here `args.images_per_forward` images' views
    form ONE network batch, and the images are sharded over the ranks (position p of the shuffled order belongs to rank
    p % world); one all-gather of the [N,K,2] locations and one of the annotations (each rank reads ONLY its own images'
    items, once: pixels and annotations together), every rank returns all of them in order.
    `dataset`: any `{"img"[, "kpts", "visibility"]}` dataset (default: `build_dataset(args)`; `synthetic` / `custom` have no
    annotations => targets None).  `draws = (order [N], noise [N*n,4,h,w], thetas [N*n,2,3])` injects the loader order
    and the per-view draws in the reference's draw order (parity tests; the reference takes them from the global RNGs)."""
    world, rank = skp_dist.world_size(), skp_dist.rank()
    dev, controller = next(iter(controllers.items()))
    if dataset is None:
        dataset = build_dataset(args)
    total = min(len(dataset), int(getattr(args, "max_num_points", 50_000)))
    n_aug = (args.augmentation_iterations // num_gpus) * num_gpus
    upscale, strategy = 512, getattr(args, "max_loc_strategy", "argmax")
    if draws is not None:
        order = [int(i) for i in draws[0]][:total]
        noise, thetas = torch.as_tensor(draws[1]), torch.as_tensor(draws[2], dtype=torch.float32)
    else:
        gen = torch.Generator().manual_seed(getattr(args, "seed", 0) + 2468)
        order, noise, thetas = torch.randperm(len(dataset), generator=gen).tolist()[:total], None, None
    mine = list(range(rank, total, world))                          # positions of the shuffled order this rank computes
    group = max(1, int(getattr(args, "images_per_forward", 4)))
    idx = torch.as_tensor(top_indices).long()
    found, targets, vis = [], [None] * total, [None] * total
    for g0 in range(0, len(mine), group):
        pos = mine[g0:g0 + group]
        items = [dataset[order[p]] for p in pos]                    # ONE load per image: pixels and annotations together
        imgs = torch.stack([torch.as_tensor(it["img"]) for it in items]).to(dev)
        for p, it in zip(pos, items):
            if "kpts" in it:
                targets[p] = torch.as_tensor(it["kpts"]).cpu()
            if "visibility" in it:
                vis[p] = torch.as_tensor(it["visibility"]).cpu()
        rows = [r for p in pos for r in range(p * n_aug, (p + 1) * n_aug)]
        maps = run_images_with_context_augmented(
            ldm, imgs, context, idx, controllers={dev: controller}, layers=args.layers,
            augmentation_iterations=args.augmentation_iterations, noise_level=args.noise_level,
            augment_degrees=args.augment_degrees, augment_scale=args.augment_scale, augment_translate=args.augment_translate,
            num_gpus=num_gpus, upscale_size=upscale, thetas=None if thetas is None else thetas[rows],
            noise=None if noise is None else noise[rows])           # [m,K,512,512]
        for i in range(len(pos)):
            found.append(keypoints_from_maps(maps[i], strategy))
    K = int(idx.numel())
    local = torch.stack(found) if found else torch.zeros(0, K, 2, device=dev)
    if world > 1:
        per = (total + world - 1) // world
        pad = torch.zeros(per, K, 2, device=dev)
        pad[:local.shape[0]] = local
        parts = [torch.empty_like(pad) for _ in range(world)]
        torch.distributed.all_gather(parts, pad)
        source = torch.stack([parts[p % world][p // world] for p in range(total)])
        # annotations travel with the locations: each rank read only its own images' items
        ann = [None] * world
        torch.distributed.all_gather_object(ann, ({p: targets[p] for p in mine if targets[p] is not None},
                                                  {p: vis[p] for p in mine if vis[p] is not None}))
        for tg, vs in ann:
            for p, v in tg.items():
                targets[p] = v
            for p, v in vs.items():
                vis[p] = v
    else:
        source = local
    have_t, have_v = all(t is not None for t in targets), all(v is not None for v in vis)
    return (source, torch.stack(targets) if have_t and total else None, torch.stack(vis) if have_v and total else None)
