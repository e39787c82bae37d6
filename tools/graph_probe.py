#!/usr/bin/env python
"""Feasibility probe: the forward + backward of one optimisation step (group_step) captured in a hipGraph (torch.cuda.graph) and
replayed, against the eager step.  Thetas / images are FIXED here (the probe measures launch cost, not training).
    python tools/graph_probe.py [--images-per-rank 1] [--steps 20]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images-per-rank", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import default_args, group_step
    from stablekeypoints_amd.optimize_token import load_ldm
    dev = torch.device("cuda", 0)
    ldm, controllers, _ = load_ldm(dev, "sd15", feature_upsample_res=128, init_on_device=True)
    controller = controllers[dev]
    n = a.images_per_rank
    args = default_args(num_tokens=77, feature_upsample_res=128, batch_size=n, device=str(dev), image_size=512)
    images = torch.rand(n, 3, 512, 512, device=dev)
    ctx = torch.randn(1, 77, 768, device=dev).requires_grad_(True)
    tr = RandomAffineWithInverse(args.augment_degrees, args.augment_scale, args.augment_translate)
    thetas = tr.sample_theta(n)
    th_dev = thetas.to(dev)

    class Fixed(RandomAffineWithInverse):                      # fixed thetas, already on the device: nothing host-side inside the capture
        def __call__(self, img, theta=None):
            import torch.nn.functional as F
            self.last_theta_host = thetas
            self.last_params = {"theta": th_dev}
            return F.grid_sample(img, F.affine_grid(th_dev, img.size(), align_corners=False), align_corners=False)
    tr = Fixed()

    def step():
        return group_step(ldm, images, ctx, args, controller, tr, denom=n, thetas=None)

    for _ in range(3):
        step(); ctx.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(); ctx.grad = None
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / a.steps * 1e3
    print(f"eager: {eager:.2f} ms / step", flush=True)
    # capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step(); ctx.grad = None
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    ctx.grad = None
    t0 = time.perf_counter()
    with torch.cuda.graph(g):
        out = step()
    print(f"captured in {time.perf_counter() - t0:.2f} s", flush=True)
    grad_static = ctx.grad
    g.replay(); torch.cuda.synchronize()
    ref = grad_static.clone()
    t0 = time.perf_counter()
    c0 = time.thread_time()
    for _ in range(a.steps):
        g.replay()
    c1 = time.thread_time() - c0
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / a.steps * 1e3
    print(f"replay: {rep:.2f} ms / step (launch thread {c1 / a.steps * 1e3:.2f} ms CPU per replay); grad identical across replays: "
          f"{bool(torch.equal(ref, grad_static))} (noise is redrawn per replay: expect False); loss {float(out[0]):.5f}")


if __name__ == "__main__":
    main()
