#!/usr/bin/env python
"""K-split sweep of the transformed-filter F(4x4,3x3) launches: time per forced split S (skp_tune_set("wino_split")) next to the
planner's own choice.   python tools/split_sweep.py [--rows 2] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import ops  # noqa: E402

SHAPES = [(512, 512, 64), (640, 640, 32), (320, 320, 64), (320, 640, 32), (640, 320, 32), (640, 1280, 16), (1280, 640, 16), (960, 320, 64), (640, 320, 64)]


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
    ap.add_argument("--rows", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    lib = ops.N.lib()
    for ci, co, s in SHAPES:
        B = a.rows
        x = torch.randn(B, ci, s, s, generator=g).cuda()
        w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).cuda()
        U = ops._wino4_filters(w, False)
        out = B * co * s * s * 4
        plan = max(1, lib.skp_conv3x3_f4_workspace(B, ci, co, s, s) // out)
        t0 = timed(lambda: ops._conv3x3_f4_raw(x, U, None, co), a.iters)
        line = f"{ci}->{co} @{s}^2 rows {B}: plan S={plan} {t0:7.1f} us |"
        for S in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            if S > ci // 16:
                continue
            ops.N.tune("wino_split", S)
            try:
                if max(1, lib.skp_conv3x3_f4_workspace(B, ci, co, s, s) // out) != S:
                    continue
                line += f" S{S}:{timed(lambda: ops._conv3x3_f4_raw(x, U, None, co), a.iters):6.1f}"
            finally:
                ops.N.tune("wino_split", 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
