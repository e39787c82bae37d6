#!/usr/bin/env python
"""The two roofline kernels of bench.py at their bench launch shapes, a few launches each and nothing else: the workload
`bench.py --traffic live` runs under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters only) to
read the HBM traffic of THIS build on THIS box.    python tools/traffic_probe.py --rows 8 --tokens 77 --res 128 --image-size 512"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    B, T, R, H, K = a.rows, a.tokens, a.res, 8, a.top_k
    ci = co = 128
    Bc = min(B, max(1, (2 ** 31 - 1) // (ci * a.image_size * a.image_size * 4)))
    x = torch.randn(Bc, ci, a.image_size, a.image_size, generator=g).to(dev)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(dev)
    U = ops._wino4_filters(w, False)
    lib, N = ops.N.lib(), ops.N
    nblk = ops.conv3x3_stats_blocks(x.shape, w.shape)
    if nblk and lib.skp_conv3x3_f4_gn_ok(Bc, ci, co, a.image_size, a.image_size):       # the form the step launches (bench.py conv_roofline)
        y = torch.empty(Bc, co, a.image_size, a.image_size, device=dev)
        stats = torch.empty(Bc, co, nblk, 2, device=dev)
        coef = torch.stack([torch.full((Bc, ci), 0.7), torch.full((Bc, ci), 0.1)], dim=-1).to(dev).contiguous()
        for _ in range(a.iters):
            N.check(lib.skp_conv3x3_f4_gn_f32(x.data_ptr(), U.data_ptr(), None, None, y.data_ptr(), stats.data_ptr(), coef.data_ptr(),
                                              Bc, ci, co, a.image_size, a.image_size, ops._stream()), "skp_conv3x3_f4_gn_f32")
        del y, stats
    else:
        for _ in range(a.iters):
            ops._conv3x3_f4_raw(x, U, None, co)
    del x
    sides = [16, 16, 16, 32]
    NT = (T + 15) // 16 * 16
    S = []
    for s in sides:
        t = torch.zeros(B, H, s * s, NT)
        t[..., :T] = torch.randn(B, H, s * s, T, generator=g) * 3
        S.append(t.to(dev))
    for _ in range(a.iters):
        M, lse = ops._map_fwd(S, sides, B, H, T, R)
    sel = torch.stack([torch.randperm(T, generator=g)[:K] for _ in range(B)]).to(dev)
    G = torch.randn(B, K, R, R, generator=g).to(dev)
    if ops.map_bwd_sparse_supported(sides, K, R, T):
        for _ in range(a.iters):
            ops._map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse)
    else:
        dM = torch.zeros(B, T, R, R, device=dev)
        for b in range(B):
            dM[b, sel[b]] = G[b]
        dS = [torch.zeros_like(t) for t in S]
        for _ in range(a.iters):
            ops._map_bwd(S, dS, sides, B, H, T, R, dM, lse)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
