#!/usr/bin/env python
"""Same-box A/B of two BUILDS of libskp_hip.so on the Winograd conv forms the step launches.

    python tools/ab_build.py stablekeypoints_amd/csrc/libskp_hip.so gpurun_out/libskp_prev.so [--rounds 3] [--shapes gn,plain,...]

Each library is timed in its own subprocess (SKP_LIB_PATH), the builds alternate A B A B ... so that clock / thermal drift
of the box does not land on one of them.  Shapes: `gn` = 128->128 @512^2 with GroupNorm folded + statistics (launches #0-#3 of
a step), `plain` = the same without either, `s256` / `s128` / `s64` = the VAE's deeper levels with the statistics epilogue,
`u320` = 320->320 @64^2, `u1280_16` / `u640_32` / `u1280_8` = K-split UNet layers, `r1280_8` / `r2560_8` / `r1280_16` /
`r2560_16` = the raw-filter form at the 8^2 / 16^2 layers."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"gn": (8, 128, 128, 512, "gn"), "plain": (8, 128, 128, 512, "plain"), "s256": (8, 256, 256, 256, "stats"),
          "s128": (8, 512, 512, 128, "stats"), "s64": (8, 512, 512, 64, "stats"), "u320": (8, 320, 320, 64, "stats"),
          "u1280_16": (8, 1280, 1280, 16, "plain"), "u640_32": (8, 640, 640, 32, "plain"), "u1280_8": (8, 1280, 1280, 8, "plain"),
          "u2560_16": (8, 2560, 1280, 16, "plain"), "u1920_32": (8, 1920, 640, 32, "plain"),
          # the raw-filter form (skp_conv3x3_f4r_f32: input transform + convolution + K-split reduction per call)
          "r1280_8": (8, 1280, 1280, 8, "raw"), "r2560_8": (8, 2560, 1280, 8, "raw"), "r1280_16": (8, 1280, 1280, 16, "raw"),
          "r2560_16": (8, 2560, 1280, 16, "raw"),
          # pN: stage-time probes (512^2, 4 rows, N input channels)
          "p128": (4, 128, 128, 512, "plain"), "p256": (4, 256, 128, 512, "plain"), "p384": (4, 384, 128, 512, "plain")}


def worker(names, iters):
    sys.path.insert(0, ROOT)
    import torch
    from stablekeypoints_amd import ops
    lib, N = ops.N.lib(), ops.N
    out = {}
    g = torch.Generator().manual_seed(0)
    for name in names:
        B, ci, co, sz, form = SHAPES[name]
        x = torch.randn(B, ci, sz, sz, generator=g).cuda()
        w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).cuda()
        y = torch.empty(B, co, sz, sz, device="cuda")
        U = ops._wino4r_filters(w, False) if form == "raw" else ops._wino4_filters(w, False)
        nblk = ops.conv3x3_stats_blocks(x.shape, w.shape)
        if form == "raw":
            fn = lambda: ops._conv3x3_f4r_raw(x, U, None, co, out=y)
        elif form == "gn":
            stats = torch.empty(B, co, nblk, 2, device="cuda")
            coef = torch.stack([torch.full((B, ci), 0.7), torch.full((B, ci), 0.1)], dim=-1).cuda().contiguous()
            fn = lambda: N.check(lib.skp_conv3x3_f4_gn_f32(x.data_ptr(), U.data_ptr(), None, None, y.data_ptr(), stats.data_ptr(),
                                                           coef.data_ptr(), B, ci, co, sz, sz, ops._stream()), "gn")
        elif form == "stats" and nblk:
            stats = torch.empty(B, co, nblk, 2, device="cuda")
            fn = lambda: ops._conv3x3_f4_raw(x, U, None, co, out=y, stats=stats)
        else:
            fn = lambda: ops._conv3x3_f4_raw(x, U, None, co, out=y)
        for _ in range(8):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / iters * 1e3
        del x, y, U, w
    print("AB_RESULT " + json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(sys.argv[2].split(","), int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default="gn,plain,s256,s128,u320,u1280_16,u640_32")
    a = ap.parse_args()
    res = {l: [] for l in a.libs}
    for _ in range(a.rounds):
        for l in a.libs:
            env = dict(os.environ, SKP_LIB_PATH=os.path.abspath(l))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", a.shapes, str(a.iters)], env=env,
                               capture_output=True, text=True)
            line = [x for x in r.stdout.splitlines() if x.startswith("AB_RESULT ")]
            if not line:
                print(l, "FAILED", r.stderr[-500:])
                continue
            res[l].append(json.loads(line[0][10:]))
    names = a.shapes.split(",")
    print("| shape | " + " | ".join(os.path.basename(l) for l in a.libs) + " |")
    print("|---|" + "---|" * len(a.libs))
    for n in names:
        cells = []
        for l in a.libs:
            v = [r[n] for r in res[l] if n in r]
            cells.append(f"{min(v):.1f} (runs {', '.join(f'{x:.0f}' for x in v)})" if v else "-")
        print(f"| {n} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
