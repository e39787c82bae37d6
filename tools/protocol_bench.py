#!/usr/bin/env python
"""BASELINE config 2 as the reference runs it: `optimize_embedding(num_steps=500, batch_size=4, num_tokens=77)` through
the public entry point (reference optimize.py:269-452; its DataLoader iteration :333-347), full-width SD-1.5 architecture at
512^2 on one MI355X, on

  (i)  the device-resident synthetic set (64 images already in HBM), and
  (ii) a HOST-resident folder of 64 PNG files written to /tmp (`dataset_name="custom"`: decode + resize + H2D per image),
       through the one-group-ahead loader (`--loader-workers`, 0 = synchronous).

Per leg, ONE JSON line: images/s over all steps, per-`--window`-step rates (clock drift), allocator bytes after step 10
and at the end, host CPU-seconds per step (whole process incl. loader threads, and the launch thread alone) next to the
wall time per step.  The only host synchronisations are at the window boundaries."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def write_png_folder(path, n, size, seed=0):
    """n RGB PNGs of smooth random fields + noise (natural-image-like compressibility: ~350 KB each at 512^2)."""
    from PIL import Image
    os.makedirs(path, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    for i in range(n):
        low = torch.rand(1, 3, size // 16, size // 16, generator=g)
        img = torch.nn.functional.interpolate(low, size=(size, size), mode="bicubic", align_corners=False)[0]
        img = (img + 0.05 * torch.randn(3, size, size, generator=g)).clamp(0, 1)
        Image.fromarray((img.permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(os.path.join(path, f"img_{i:03d}.png"))
    return path


def run_leg(name, ldm, controllers, args, window):
    from stablekeypoints_amd.optimize import optimize_embedding
    dev = next(iter(controllers))
    marks, mem = [], {}
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()

    def cb(step):
        if step + 1 == 10:
            mem["allocated_step10"], mem["reserved_step10"] = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
        if (step + 1) % window == 0 or step + 1 == args.num_steps:
            torch.cuda.synchronize()
            marks.append((step + 1, time.perf_counter(), time.process_time(), time.thread_time()))

    t0, c0, th0 = time.perf_counter(), time.process_time(), time.thread_time()
    ctx = optimize_embedding(ldm, args, controllers, 1, step_callback=cb)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    steps, bs = int(args.num_steps), int(args.batch_size)
    rates, prev = [], (0, t0, c0, th0)
    for m in marks:
        ds, dt = m[0] - prev[0], m[1] - prev[1]
        rates.append({"steps": [prev[0] + 1, m[0]], "images_per_s": bs * ds / dt, "ms_per_step": dt / ds * 1e3,
                      "host_cpu_s_per_step_process": (m[2] - prev[2]) / ds, "host_cpu_s_per_step_launch_thread": (m[3] - prev[3]) / ds})
        prev = m
    steady = [r for r in rates[1:]] or rates
    return {"leg": name, "entry_point": "stablekeypoints_amd.optimize.optimize_embedding", "num_steps": steps, "batch_size": bs,
            "num_tokens": int(args.num_tokens), "image_size": int(args.image_size), "dataset": args.dataset_name,
            "loader_workers": int(getattr(args, "loader_workers", 0)),
            "images_per_s_all_steps": bs * steps / (t1 - t0), "ms_per_step_all_steps": (t1 - t0) / steps * 1e3,
            "images_per_s_after_first_window": bs * sum(r["steps"][1] - r["steps"][0] + 1 for r in steady)
            / sum((r["steps"][1] - r["steps"][0] + 1) * r["ms_per_step"] * 1e-3 for r in steady),
            "windows": rates, "allocator": {**mem, "allocated_end": torch.cuda.memory_allocated(),
                                            "reserved_end": torch.cuda.memory_reserved(), "peak_allocated": torch.cuda.max_memory_allocated()},
            "host_cpu_s_per_step_process": (time.process_time() - c0) / steps,
            "host_cpu_s_per_step_launch_thread": (time.thread_time() - th0) / steps,
            "embedding_abs_sum": float(ctx.double().abs().sum()), "device": torch.cuda.get_device_name(0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--model", default="sd15")
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--candidates", type=int, default=25)
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--n-images", type=int, default=64)
    ap.add_argument("--legs", default="device,host", help="comma list of device, host, host-sync")
    ap.add_argument("--loader-workers", type=int, default=4)
    ap.add_argument("--folder", default="/tmp/skp_protocol_images")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    assert torch.cuda.is_available(), "protocol_bench.py measures the MI355X path"
    from stablekeypoints_amd import _native, tuning
    from stablekeypoints_amd.optimize import default_args
    from stablekeypoints_amd.optimize_token import load_ldm
    _native.lib()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ldm, controllers, _ = load_ldm(dev, a.model, feature_upsample_res=a.res, init_on_device=True)
    tuning.enable()
    common = dict(num_steps=a.steps, batch_size=a.batch_size, num_tokens=a.tokens, feature_upsample_res=a.res, top_k=a.top_k,
                  furthest_point_num_samples=a.candidates, image_size=a.image_size, device=str(dev), log_interval=0, seed=0)
    lines = []
    for leg in a.legs.split(","):
        if leg == "device":
            args = default_args(dataset_name="synthetic", max_len=a.n_images, **common)
        elif leg in ("host", "host-sync"):
            if not os.path.isdir(a.folder) or len(os.listdir(a.folder)) != a.n_images:
                write_png_folder(a.folder, a.n_images, a.image_size)
            args = default_args(dataset_name="custom", dataset_loc=a.folder,
                                loader_workers=a.loader_workers if leg == "host" else 0, **common)
        else:
            raise SystemExit(f"unknown leg {leg}")
        # a short untimed run first: solver selection, allocator growth, first-touch of the loader
        warm = default_args(**{**vars(args), "num_steps": 3})
        from stablekeypoints_amd.optimize import optimize_embedding
        optimize_embedding(ldm, warm, controllers, 1)
        line = run_leg(leg, ldm, controllers, args, a.window)
        print(json.dumps(line), flush=True)
        lines.append(line)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(lines, f, indent=1)


if __name__ == "__main__":
    main()
