#!/usr/bin/env python
"""A/B timing + cross-check of the flash-attention kernel generations / tile variants at the step's launch shapes.
    python tools/fa_bench.py [--rows 8] [--iters 10] [--bwd]
Variants: "0" = the library's choice, "0+2k" = the two-kernel backward at the fused form's shapes (skp_tune_set)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import ops  # noqa: E402

PEAK = 157.3


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--variants", default="0,0+2k")
    ap.add_argument("--shapes", default="4096x8x40,1024x8x80,9216x5x64,4096x10x64")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    for shp in a.shapes.split(","):
        N, H, d = (int(x) for x in shp.split("x"))
        B = a.rows
        q, k, v, w = (torch.randn(B, N, H * d, generator=g).to(dev) for _ in range(4))
        q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
        flops = 4.0 * N * N * d * B * H
        ref = gref = None
        for var in a.variants.split(","):
            ops.N.tune("fa2_two_kernel_bwd", 1 if var.endswith("+2k") else 0)
            try:
                out = ops.self_attention(q, k, v, H, d ** -0.5)
                torch.cuda.synchronize()
            except RuntimeError as e:
                print(f"N={N} H={H} d={d} variant {var}: {e}")
                continue
            t = timed(lambda: ops.self_attention(q, k, v, H, d ** -0.5), a.iters)
            line = f"N={N} H={H} d={d} B={B} variant {var:>5}: fwd {t * 1e3:8.3f} ms  {flops / t / 1e12:6.1f} TF/s ({flops / t / 1e12 / PEAK:.3f})"
            if ref is None:
                ref = out.detach().clone()
            else:
                line += f"  max|diff vs first| {float((out.detach() - ref).abs().max()):.2e}"
            if a.bwd:
                o = ops.self_attention(q, k, v, H, d ** -0.5)
                tb = timed(lambda: torch.autograd.grad(o, (q, k, v), w, retain_graph=True), max(3, a.iters // 2))
                line += f" | bwd {tb * 1e3:8.3f} ms {2.5 * flops / tb / 1e12:6.1f} TF/s ({2.5 * flops / tb / 1e12 / PEAK:.3f})"
                gr = [x.clone() for x in torch.autograd.grad(o, (q, k, v), w, retain_graph=True)]
                if gref is None:
                    gref = gr
                else:
                    line += "  grad diff " + "/".join(f"{float((x - y).abs().max()):.1e}" for x, y in zip(gr, gref))
            print(line, flush=True)


if __name__ == "__main__":
    main()
