#!/usr/bin/env python
"""Split-bf16 GEMM (experiment) vs the library fp32 GEMM at the step's nn.Linear shapes: time, TF/s-equivalent, max error vs fp64.
    python tools/gemm_x3_bench.py [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import ops  # noqa: E402

SHAPES = [(32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640), (8192, 640, 2560),
          (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (512, 1280, 1280)]


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    for M, N, K in SHAPES:
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        ref = torch.nn.functional.linear(x[:256].double(), w.double(), b.double())
        y32 = torch.nn.functional.linear(x, w, b)
        y3 = ops.LinearX3Fn.apply(x, w, b)
        t32 = timed(lambda: torch.nn.functional.linear(x, w, b), a.iters)
        planes = ops._x3_planes(w, False)
        y = torch.empty(M, N, device=x.device)
        st = torch.cuda.current_stream().cuda_stream
        lib = ops.N.lab()
        t3 = timed(lambda: lib.skp_gemm_x3_nt_f32(x.data_ptr(), planes.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, K, N, st),
                   a.iters)                                     # straight through the C-ABI: no per-call Python allocation
        fl = 2.0 * M * N * K
        print(f"M={M:6d} N={N:6d} K={K:5d}: fp32 lib {t32 * 1e6:8.1f} us {fl / t32 / 1e12:6.1f} TF/s err {float((y32[:256].double() - ref).abs().max()):.2e}"
              f" | bf16x3 {t3 * 1e6:8.1f} us {fl / t3 / 1e12:6.1f} TF/s-eq err {float((y3[:256].double() - ref).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
