#!/usr/bin/env python
"""Repeatability soak of the round-6 attention forms (token-split cross-attention, flash backward range splits): 300 / 100 repeats per
shape on the same inputs must be bit-identical (a race between waves or an unordered sum would show).  python tools/attn_repeat_soak.py"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stablekeypoints_amd import ops
g = torch.Generator().manual_seed(3)
bad = 0
for (B, N, H, d, T) in [(8, 256, 8, 160, 77), (2, 1024, 8, 80, 77), (8, 64, 8, 160, 77), (2, 250, 4, 160, 100), (2, 256, 8, 160, 77)]:
    C = H * d
    q = torch.randn(B, N, C, generator=g).cuda().requires_grad_(True)
    k = torch.randn(1, T, C, generator=g).cuda().requires_grad_(True)
    v = torch.randn(1, T, C, generator=g).cuda().requires_grad_(True)
    w = torch.randn(B, N, C, generator=g).cuda()
    ref = None
    for it in range(300):
        o = ops.cross_attention(q, k, v, H, d ** -0.5)
        gr = torch.autograd.grad(o, (q, k, v), w)
        cur = [o.detach()] + [x for x in gr]
        if ref is None:
            ref = [x.clone() for x in cur]
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            bad += 1
    # self-attention d=160 / d=80 backward splits
    x = [torch.randn(B, N, C, generator=g).cuda().requires_grad_(True) for _ in range(3)]
    ref = None
    for it in range(100):
        o = ops.self_attention(x[0], x[1], x[2], H, d ** -0.5)
        gr = torch.autograd.grad(o, x, w)
        cur = [o.detach()] + list(gr)
        if ref is None:
            ref = [t.clone() for t in cur]
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            bad += 1
    print((B, N, H, d, T), "mismatching repeats so far:", bad, flush=True)
print("SOAK", "OK" if bad == 0 else "FAIL", bad)
