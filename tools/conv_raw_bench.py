#!/usr/bin/env python
"""Small-spatial 3x3 convolutions: the raw-filter form (skp_conv3x3_f4r_f32) against the transformed-filter F(4x4,3x3) kernels,
us per call (input transform, K-split reduction included) at the UNet's 8^2 / 16^2 / 32^2 shapes of the step (8 rows).
    python tools/conv_raw_bench.py [--iters 20]          (--max-tiles 512 lets the 32^2 shapes through the gate: skp_tune_set("wino_raw_max_tiles"))"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import ops  # noqa: E402

SHAPES = [(8, 1280, 1280, 8), (8, 2560, 1280, 8), (8, 1280, 2560, 8), (8, 1280, 1280, 16), (8, 2560, 1280, 16), (8, 1280, 2560, 16),
          (8, 1920, 1280, 16), (8, 1280, 1920, 16), (8, 640, 1280, 16), (8, 1280, 640, 16),
          (8, 640, 640, 32), (8, 1280, 1280, 32), (8, 1280, 640, 32), (8, 1920, 640, 32), (8, 640, 1920, 32), (8, 960, 640, 32),
          (8, 512, 512, 64), (8, 320, 320, 64), (8, 640, 320, 64), (8, 960, 320, 64)]        # far outside the gate (--max-tiles 4096)


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rows", type=int, default=0, help="batch rows instead of the step's 8 (the augmented inference runs 10 / 20 / 40)")
    ap.add_argument("--max-tiles", type=int, default=0, help="widen the raw-filter form's gate to this many tiles")
    ap.add_argument("--only", default="", help='substring filter on the shape, e.g. "1280->1280 @8"')
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    lib = ops.N.lib()
    if a.max_tiles:
        ops.N.tune("wino_raw_max_tiles", a.max_tiles)
    print("| shape | F(4x4,3x3), transformed filter (us) | raw-filter form (us) | frac of 157.3 TF/s | max diff |")
    print("|---|---|---|---|---|")
    only = a.only
    for B, ci, co, s in SHAPES:
        if only and only not in f"{ci}->{co} @{s}^2":
            continue
        B = a.rows or B
        x = torch.randn(B, ci, s, s, generator=g).cuda()
        w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).cuda()
        U = ops._wino4_filters(w, False)
        t_old = timed(lambda: ops._conv3x3_f4_raw(x, U, None, co), a.iters)
        if not lib.skp_conv3x3_f4r_ok(B, ci, co, s, s):
            print(f"| {ci}->{co} @{s}^2 | {t_old:.1f} | (gate closed) | | |")
            continue
        R = ops._wino4r_filters(w, False)
        t_new = timed(lambda: ops._conv3x3_f4r_raw(x, R, None, co), a.iters)
        d = (ops._conv3x3_f4r_raw(x, R, None, co) - ops._conv3x3_f4_raw(x, U, None, co)).abs().max().item()
        fl = 2.0 * 9 * ci * co * B * s * s / 4
        print(f"| {ci}->{co} @{s}^2 | {t_old:.1f} | {t_new:.1f} | {fl / t_new / 1e6 / 157.3:.3f} | {d:.2e} |", flush=True)
        del x, w, U, R


if __name__ == "__main__":
    main()
