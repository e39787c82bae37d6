#!/usr/bin/env python
"""Winograd conv3x3 kernel vs the library convolution: error and time per shape (fwd and bwd-data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stablekeypoints_amd import ops

SHAPES = [  # name, B, Cin, Cout, H
    ("vae 512^2 128->128", 8, 128, 128, 512), ("vae 256^2 128->256", 8, 128, 256, 256),
    ("vae 256^2 256->256", 8, 256, 256, 256), ("vae 128^2 256->512", 8, 256, 512, 128),
    ("vae 128^2 512->512", 8, 512, 512, 128), ("vae 64^2 512->512", 8, 512, 512, 64),
    ("unet 64^2 320->320", 8, 320, 320, 64), ("unet 64^2 640->320", 8, 640, 320, 64), ("unet 64^2 960->320", 8, 960, 320, 64), ("unet 32^2 320->640", 8, 320, 640, 32),
    ("unet 32^2 640->640", 8, 640, 640, 32), ("unet 16^2 640->1280", 8, 640, 1280, 16),
    ("unet 16^2 1280->1280", 8, 1280, 1280, 16), ("unet 8^2 1280->1280", 8, 1280, 1280, 8), ("unet 8^2 2560->1280", 8, 2560, 1280, 8),
    ("unet 16^2 2560->1280", 8, 2560, 1280, 16), ("unet 32^2 1920->640", 8, 1920, 640, 32),
    ("unet 32^2 1280->640", 8, 1280, 640, 32), ("unet 32^2 960->640", 8, 960, 640, 32),
    # stage-time probes: same spatial size and output channels, growing Cin (SKP_BENCH_ONLY=probe)
    ("probe 512^2 B4 128->128", 4, 128, 128, 512), ("probe 512^2 B4 256->128", 4, 256, 128, 512), ("probe 512^2 B4 384->128", 4, 384, 128, 512),
    ("probe 256^2 B8 128->128", 8, 128, 128, 256), ("probe 256^2 B8 256->128", 8, 256, 128, 256), ("probe 256^2 B8 512->128", 8, 512, 128, 256),
    ("probe 128^2 B8 128->512", 8, 128, 512, 128), ("probe 128^2 B8 256->512", 8, 256, 512, 128), ("probe 128^2 B8 512->512", 8, 512, 512, 128),
    # backward-data launches (channel counts swapped)
    ("bwd 16^2 1280->2560", 8, 1280, 2560, 16), ("bwd 8^2 1280->2560", 8, 1280, 2560, 8), ("bwd 32^2 640->1920", 8, 640, 1920, 32),
    ("bwd 16^2 1280->1920", 8, 1280, 1920, 16), ("bwd 32^2 640->1280", 8, 640, 1280, 32), ("bwd 16^2 1280->640", 8, 1280, 640, 16),
    ("bwd 32^2 640->960", 8, 640, 960, 32), ("bwd 64^2 320->640", 8, 320, 640, 64), ("bwd 64^2 320->960", 8, 320, 960, 64),
]


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    variants = [int(v) for v in sys.argv[1:]] or [0]
    torch.manual_seed(0)
    only = os.environ.get("SKP_BENCH_ONLY", "")              # substring filter on the shape names
    for name, B, ci, co, H in SHAPES:
        if only and not any(o in name for o in only.split(",")):
            continue
        x = torch.randn(B, ci, H, H, device="cuda")
        w = torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)
        b = torch.randn(co, device="cuda")
        ref = F.conv2d(x, w, b, padding=1)
        if os.environ.get("SKP_BENCH_F4_ONLY"):                    # the F(4x4) launch alone (K-split plan A/Bs)
            U4 = ops._wino4_filters(w, False)
            t4 = timeit(lambda: ops._conv3x3_f4_raw(x, U4, b, co), 30)
            err4 = ((ops._conv3x3_f4_raw(x, U4, b, co) - ref).abs().max() / ref.abs().max()).item()
            ws = ops.N.lib().skp_conv3x3_f4_workspace(B, ci, co, H, H)
            nblk = ops.conv3x3_stats_blocks(x.shape, w.shape)
            ts = float("nan")
            if nblk:                                              # the same launch with the block-statistics epilogue
                st = torch.empty(B, co, nblk, 2, device="cuda")
                ts = timeit(lambda: ops._conv3x3_f4_raw(x, U4, b, co, stats=st), 30)
            print(f"{name:24s} F4 {t4 * 1e3:7.1f} us splits {ws // (B * co * H * H * 4) if ws else 1:2d} err {err4:.1e} | with statistics "
                  f"{ts * 1e3:7.1f} us", flush=True)
            continue
        U = ops._wino_filters(w, False)
        fl = 2 * 9 * ci * co * B * H * H / 1e9
        t_lib = timeit(lambda: F.conv2d(x, w, b, padding=1))
        line = f"{name:24s} lib {t_lib:6.2f} ms {fl / t_lib:6.1f} TF/s |"
        for v in variants:
            for split in (False, True):
                if split and not ops.N.lib().skp_conv3x3_workspace(B, ci, co, H, H, v):
                    continue
                y = ops._conv3x3_raw(x, U, b, co, v, split)
                err = ((y - ref).abs().max() / ref.abs().max()).item()
                t = timeit(lambda: ops._conv3x3_raw(x, U, b, co, v, split))
                line += f" v{v}{'s' if split else ' '} {t:6.2f} ms {fl / t:6.1f} TF/s-eq err {err:.1e} |"
        if H % 4 == 0:
            U4 = ops._wino4_filters(w, False)
            y4 = ops._conv3x3_f4_raw(x, U4, b, co)
            err4 = ((y4 - ref).abs().max() / ref.abs().max()).item()
            t4 = timeit(lambda: ops._conv3x3_f4_raw(x, U4, b, co))
            line += f" F4 {t4:6.2f} ms {fl / t4:6.1f} TF/s-eq err {err4:.1e} |"
            del U4, y4
        # backward-data
        dy = torch.randn_like(ref)
        Ub = ops._wino_filters(w, True)
        dref = torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1)
        dx = ops._conv3x3_raw(dy, Ub, None, ci, 0)
        errb = ((dx - dref).abs().max() / dref.abs().max()).item()
        tb_lib = timeit(lambda: torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1))
        tb = timeit(lambda: ops._conv3x3_raw(dy, Ub, None, ci, 0))
        line += f" bwd lib {tb_lib:6.2f} ms wino {tb:6.2f} ms err {errb:.1e}"
        print(line, flush=True)
        del x, w, ref, U, Ub, dy, dref, dx


if __name__ == "__main__":
    main()
