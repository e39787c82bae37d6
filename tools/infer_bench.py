#!/usr/bin/env python
"""Augmented inference (eval.run_image_with_context_augmented, reference eval.py:197-355): seconds per image for n affine views of a
512^2 image through the early-exit UNet forward, the K selected maps and the fused un-warp tail; and the arg-max keypoints.
    python tools/infer_bench.py [--tokens 77] [--views 10] [--iters 10] [--profile]
With --dataset N: the dataset-level driver keypoint_regressor.precompute_all_keypoints over N synthetic 512^2 images (reference
keypoint_regressor.py:111-198: loop -> augmented inference -> map -> location), several images' views per network batch, in
images/s for --group 1, 2, 4 images per forward."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--views", type=int, default=10)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--dataset", type=int, default=0, help="images of the dataset-level driver run (0: single-image timing only)")
    ap.add_argument("--group", default="1,2,4", help="images per network batch to try with --dataset")
    a = ap.parse_args()
    from stablekeypoints_amd.eval import find_max_pixel, run_image_with_context_augmented
    from stablekeypoints_amd.optimize_token import load_ldm
    dev = torch.device("cuda", 0)
    ldm, controllers, n = load_ldm(dev, "sd15", feature_upsample_res=128, init_on_device=True)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(1, a.tokens, 768, generator=g).to(dev)
    idx = torch.randperm(a.tokens, generator=g)[:a.top_k]
    img = torch.rand(3, 512, 512, generator=g).to(dev)

    def one():
        with torch.no_grad():
            maps = run_image_with_context_augmented(ldm, img, ctx, idx, device=dev, controllers=controllers, num_gpus=n,
                                                    augmentation_iterations=a.views, upscale_size=512)
            return find_max_pixel(maps)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        kp = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"T={a.tokens} views={a.views} K={a.top_k}: {dt * 1e3:.1f} ms per image ({a.views / dt:.1f} views/s), keypoints {tuple(kp.shape)}")
    if a.dataset:
        from stablekeypoints_amd.keypoint_regressor import precompute_all_keypoints
        from stablekeypoints_amd.optimize import SyntheticImages, default_args
        data = SyntheticImages(n=a.dataset, size=512, seed=3, device=dev)
        for grp in [int(v) for v in a.group.split(",")]:
            args = default_args(num_tokens=a.tokens, top_k=a.top_k, augmentation_iterations=a.views, max_num_points=a.dataset,
                                images_per_forward=grp, max_loc_strategy="argmax", device=str(dev), augment_degrees=30,
                                augment_scale=(0.9, 1.1), augment_translate=(0.1, 0.1))
            precompute_all_keypoints(ldm, ctx, idx, args, controllers, n, dataset=data)      # warm (allocator, first-use filters)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            src, _, _ = precompute_all_keypoints(ldm, ctx, idx, args, controllers, n, dataset=data)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"dataset driver: {a.dataset} images x {a.views} views, {grp} image(s) per forward: {a.dataset / dt:.2f} images/s "
                  f"({dt / a.dataset * 1e3:.1f} ms per image), source keypoints {tuple(src.shape)}")
    if a.profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            one(); torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))


if __name__ == "__main__":
    main()
