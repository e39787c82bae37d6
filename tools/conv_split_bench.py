#!/usr/bin/env python
"""Split (3 x bf16) Winograd convolution (csrc/skp_conv_wino4s.hip) against the fp32-instruction kernels: error vs fp64 and time per
shape.  `python tools/conv_split_bench.py [--json out.json] [--only substr]`.  Errors are max |y - y64| / max |y64| with the
fp64 reference computed by the library on the same device (chunked); times are events on the launch stream, interleaved
fp32 / split rounds, median of rounds."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stablekeypoints_amd import ops

SHAPES = [  # name, B, Cin, Cout, H, input kind
    ("vae 512^2 128->128", 8, 128, 128, 512, "randn"), ("vae 512^2 128->128 B2", 2, 128, 128, 512, "randn"),
    ("vae 256^2 128->256", 8, 128, 256, 256, "randn"), ("vae 256^2 256->256", 8, 256, 256, 256, "randn"),
    ("vae 128^2 256->512", 8, 256, 512, 128, "randn"), ("vae 128^2 512->512", 8, 512, 512, 128, "randn"),
    ("vae 64^2 512->512", 8, 512, 512, 64, "randn"), ("unet 64^2 320->320", 8, 320, 320, 64, "randn"),
    ("unet 64^2 640->320", 8, 640, 320, 64, "randn"), ("unet 32^2 640->640", 8, 640, 640, 32, "randn"),
    ("unet 32^2 1280->640", 8, 1280, 640, 32, "randn"), ("unet 16^2 1280->1280", 8, 1280, 1280, 16, "randn"),
    ("unet 8^2 1280->1280", 8, 1280, 1280, 8, "randn"),
    ("stat 128^2 128->128 mean50", 2, 128, 128, 128, "mean50"), ("stat 128^2 128->128 x1e-4", 2, 128, 128, 128, "tiny"),
    ("stat 128^2 128->128 x1e4", 2, 128, 128, 128, "huge"), ("stat 128^2 128->128 silu", 2, 128, 128, 128, "silu"),
]


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def ref64(x, w, b):
    out = []
    for i in range(x.shape[0]):                                     # one image at a time: fp64 conv workspace
        out.append(F.conv2d(x[i:i + 1].double(), w.double(), b.double(), padding=1))
    return torch.cat(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    torch.manual_seed(0)
    rows = []
    for name, B, ci, co, H, kind in SHAPES:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        x = torch.randn(B, ci, H, H, device="cuda")
        if kind == "mean50": x = 50.0 + 0.1 * x
        elif kind == "tiny": x = x * 1e-4
        elif kind == "huge": x = x * 1e4
        elif kind == "silu": x = F.silu(x)
        w = torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)
        b = torch.randn(co, device="cuda") * (x.abs().mean().item())
        r = ref64(x, w, b)
        scale = r.abs().max().item()
        U4, Us = ops._wino4_filters(w, False), ops._wino4s_filters(w, False)
        raw = ops.conv3x3_f4r_ok(x.shape, co)
        f32 = (lambda: ops._conv3x3_f4r_raw(x, ops._wino4r_filters(w, False), b, co)) if raw else (lambda: ops._conv3x3_f4_raw(x, U4, b, co))
        spl = lambda: ops._conv3x3_f4s_raw(x, Us, b, co)
        e32 = (f32().double() - r).abs().max().item() / scale
        esp = (spl().double() - r).abs().max().item() / scale
        del r
        for _ in range(3):
            f32(); spl()
        t32, tsp = [], []
        iters = 10 if H >= 256 else 30
        for _ in range(a.rounds):
            t32.append(timeit(f32, iters)); tsp.append(timeit(spl, iters))
        t32.sort(); tsp.sort()
        m32, msp = t32[len(t32) // 2], tsp[len(tsp) // 2]
        gf = 2 * 9 * ci * co * B * H * H / 4 / 1e6               # Winograd-domain MFLOP (direct / 4): MFLOP / us = TFLOP/s
        S = ops.N.lib().skp_conv3x3_f4s_workspace(B, ci, co, H, H) // (B * co * H * H * 4) or 1
        row = dict(shape=name, B=B, Cin=ci, Cout=co, H=H, input=kind, f32_kernel="f4r" if raw else "f4", f32_us=m32, split_us=msp,
                   speedup=m32 / msp, f32_err=e32, split_err=esp, err_ratio=esp / e32, split_ksplits=int(S),
                   f32_tf=gf / m32, split_tf_equiv=gf / msp)
        rows.append(row)
        print(f"{name:30s} f32({row['f32_kernel']:3s}) {m32:8.1f} us {row['f32_tf']:6.1f} TF/s | split {msp:8.1f} us {row['split_tf_equiv']:6.1f} TF/s-eq "
              f"S={S:2d} | x{row['speedup']:.2f} | err f32 {e32:.2e} split {esp:.2e} ({row['err_ratio']:.2f}x)", flush=True)
        del x, w, U4, Us
        torch.cuda.empty_cache()
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
