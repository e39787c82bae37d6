#!/usr/bin/env python
"""Micro-benchmark of the north-star map kernels at the step's launch shape (SD-1.5 hooked layers 3 x 16^2 + 32^2, H = 8,
R = 128): forward, dense-gradient backward (two kernels + dV staging) and sparse-gradient token-major backward.
    python tools/map_bench.py [--rows 8] [--tokens 77] [--top-k 10] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import ops  # noqa: E402


def timed(name, fn, iters, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:44s} {us:9.1f} us", flush=True)
    return us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sides", default="16,16,16,32")
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--skip-dense", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    B, T, R, H, K = a.rows, a.tokens, a.res, a.heads, a.top_k
    sides = [int(v) for v in a.sides.split(",")]
    NT = (T + 15) // 16 * 16
    S = []
    for s in sides:
        x = torch.zeros(B, H, s * s, NT)
        x[..., :T] = torch.randn(B, H, s * s, T, generator=g) * 3
        S.append(x.to(dev))
    M, lse = ops._map_fwd(S, sides, B, H, T, R)
    sel = torch.stack([torch.randperm(T, generator=g)[:K] for _ in range(B)]).to(dev)
    G = torch.randn(B, K, R, R, generator=g).to(dev)
    dM = torch.zeros(B, T, R, R, device=dev)
    for b in range(B):
        dM[b, sel[b]] = G[b]
    print(f"B={B} T={T} R={R} H={H} sides={sides} K={K}")
    timed("map forward", lambda: ops._map_fwd(S, sides, B, H, T, R), a.iters)
    if T > 128 and ops.map_wide_supported(T, R):
        ops.MAP_WIDE = False
        timed("map forward, token groups x two passes", lambda: ops._map_fwd(S, sides, B, H, T, R), a.iters)
        ops.MAP_WIDE = True
        tokrow = torch.full((T,), -1, dtype=torch.int32)
        tokrow[sel[0].cpu()] = torch.arange(K, dtype=torch.int32)
        tokrow = tokrow.to(dev)
        timed(f"map forward, {K} rows written", lambda: ops._map_fwd(S, sides, B, H, T, R, tokrow=tokrow, n_rows=K), a.iters)
    if not a.skip_dense:
        dD = [torch.zeros_like(x) for x in S]
        timed("map backward, dense gradient (A + B)", lambda: ops._map_bwd(S, dD, sides, B, H, T, R, dM, lse), a.iters)
    if ops.map_bwd_col_supported(sides, K, R, T, H):
        ops.MAP_BWD_MODE = "col"
        timed("map backward, sparse gradient (column sweep)", lambda: ops._map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse), a.iters)
    ops.MAP_BWD_MODE = "sweep"
    timed("map backward, sparse gradient (token-major)", lambda: ops._map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse), a.iters)
    if not a.skip_dense:
        dS = ops._map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse)
        print("max rel diff sparse vs dense:", max(float((x[..., :T] - y[..., :T]).abs().max() / y.abs().max()) for x, y in zip(dS, dD)))


if __name__ == "__main__":
    main()
