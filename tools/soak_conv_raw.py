"""Soak of the raw-filter convolution form (LDS DMA staging, K-split partials): repeated calls must stay bit-identical while
other allocations move its workspace around.   python tools/soak_conv_raw.py"""
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stablekeypoints_amd import ops
g = torch.Generator().manual_seed(5)
bad = 0
for (B, ci, co, s) in [(8, 1280, 1280, 8), (8, 1280, 1280, 16), (8, 1920, 640, 32), (8, 2560, 1280, 8)]:
    x = torch.randn(B, ci, s, s, generator=g).cuda()
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).cuda()
    r = torch.randn(B, co, s, s, generator=g).cuda()
    b = torch.randn(co, generator=g).cuda()
    R = ops._wino4r_filters(w, False)
    y0 = ops._conv3x3_f4r_raw(x, R, b, co, residual=r).clone()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double()
    err = (y0.double() - ref).abs().max().item() / ref.abs().max().item()
    n = 1500
    for i in range(n):
        y = ops._conv3x3_f4r_raw(x, R, b, co, residual=r)
        if i % 50 == 0:      # interleave other work so that the workspace moves around in the allocator
            tmp = torch.randn(1 << 20, device="cuda")
        if not torch.equal(y, y0):
            bad += 1
    torch.cuda.synchronize()
    print(f"{ci}->{co} @{s}^2: rel err vs fp64 {err:.2e}, {n} repeats, mismatches so far {bad}", flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED")
