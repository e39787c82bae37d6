#!/usr/bin/env python
"""EXPERIMENT: does running the NEXT optimizer step's VAE encode on a second HIP stream under the current step's UNet
forward / backward buy anything?  (The VAE input does not depend on the embedding, so the reference's data pipeline
could legally prefetch it.)  Prints ms/step for the sequential loop and for the pipelined one.
    python tools/overlap_probe.py [--steps 10]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import dist as D, ptp_utils  # noqa: E402
from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse  # noqa: E402
from stablekeypoints_amd.optimize import SyntheticImages, default_args, group_step  # noqa: E402
from stablekeypoints_amd.optimize_token import load_ldm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ldm, controllers, _ = load_ldm(dev, "sd15", feature_upsample_res=128, init_on_device=True)
    controller = controllers[dev]
    n = 4
    args = default_args(num_tokens=77, feature_upsample_res=128, batch_size=n, device=str(dev), image_size=512)
    data = SyntheticImages(n=16, size=512, seed=0, device=dev)
    ctx = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(True)
    opt = torch.optim.Adam([ctx], lr=args.lr)
    reducer = D.EmbeddingReducer(ctx, opt)
    tr = RandomAffineWithInverse(args.augment_degrees, args.augment_scale, args.augment_translate)
    cursor = [0]

    def batch():
        idx = [(cursor[0] + i) % len(data) for i in range(n)]
        cursor[0] += n
        return torch.stack([data[i]["img"] for i in idx])

    def sequential():
        group_step(ldm, batch(), ctx, args, controller, tr, denom=n)
        reducer.step()

    side = torch.cuda.Stream()
    real_i2l = ptp_utils.image2latent
    pending = {}

    def prefetch():
        images = batch()
        thetas = tr.sample_theta(n)
        ev = torch.cuda.Event()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            warped = RandomAffineWithInverse()(images, theta=thetas)
            lat = real_i2l(ldm, torch.cat([images, warped]), dev)
            ev.record(side)
        return images, thetas, lat, ev

    def pipelined(state):
        images, thetas, lat, ev = state
        nxt = prefetch()                                           # next step's VAE goes to the side stream first
        torch.cuda.current_stream().wait_event(ev)
        ptp_utils.image2latent = lambda model, image, device: lat
        try:
            group_step(ldm, images, ctx, args, controller, tr, denom=n, thetas=thetas)
        finally:
            ptp_utils.image2latent = real_i2l
        reducer.step()
        return nxt

    for mode in ("sequential", "pipelined", "sequential", "pipelined"):
        state = prefetch() if mode == "pipelined" else None
        for _ in range(a.warmup):
            state = pipelined(state) if mode == "pipelined" else sequential()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            state = pipelined(state) if mode == "pipelined" else sequential()
        torch.cuda.synchronize()
        print(f"{mode:11s} {(time.perf_counter() - t0) / a.steps * 1e3:8.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
