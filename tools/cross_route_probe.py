#!/usr/bin/env python
"""Cross-attention at T <= 128: the fused kernel (skp_cross_attn_*) against the flash kernels on the same inputs, per layer shape.
    rocprofv3 --kernel-trace --stats -- python tools/cross_route_probe.py {cross|cross128|flash} [rows]     (kernel time = sum of the stats)
    python tools/cross_route_probe.py check [rows]                                                  (max differences between the two)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import ops  # noqa: E402

SHAPES = [(4096, 8, 40), (1024, 8, 80), (256, 8, 160), (64, 8, 160)]


def main():
    which = sys.argv[1]
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    only = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
    g = torch.Generator().manual_seed(0)
    for (Nq, H, d) in SHAPES:
        if only and Nq not in only:
            continue
        C, T = H * d, 77
        q = torch.randn(rows, Nq, C, generator=g).cuda().requires_grad_(True)
        k = torch.randn(1, T, C, generator=g).cuda().requires_grad_(True)
        v = torch.randn(1, T, C, generator=g).cuda().requires_grad_(True)
        w = torch.randn(rows, Nq, C, generator=g).cuda()
        fns = {"cross": ops.CrossAttnFn, "cross128": ops.CrossAttnFn, "flash": ops.FlashAttnFn}
        ops.N.tune("cross_attn_ts", 1 if which == "cross128" else 0)     # cross128: the 128-query kernels where the token-split form would run
        if which == "check":
            res = {}
            for name, fn in fns.items():
                o = fn.apply(q, k, v, H, d ** -0.5)
                res[name] = [o.detach()] + list(torch.autograd.grad(o, (q, k, v), w))
            print(Nq, d, " ".join(f"{float((a - b).abs().max() / b.abs().max()):.1e}" for a, b in zip(res["flash"], res["cross"])))
            continue
        fn = fns[which]
        for _ in range(20):
            o = fn.apply(q, k, v, H, d ** -0.5)
            torch.autograd.grad(o, (q, k, v), w)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
