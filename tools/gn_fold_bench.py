#!/usr/bin/env python
"""GroupNorm + SiLU folded into the Winograd patch load vs the separate apply pass, per launch shape (us; events).
ops.N.tune("gn_fold_max_cout", n) widens the fold's gate beyond one channel group (the SiLU is then redone per channel group)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablekeypoints_amd import ops

def timeit(fn, iters=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (B, C, co, S) in ((8, 128, 128, 512), (8, 256, 256, 256), (8, 512, 512, 128), (8, 512, 512, 64)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, S, S, generator=g).cuda()
    w = (torch.randn(co, C, 3, 3, generator=g) / (3 * C ** 0.5)).cuda()
    norm = torch.nn.GroupNorm(32, C, eps=1e-6).cuda()
    with torch.no_grad():
        sep = lambda: ops.conv3x3_auto(ops.group_norm_silu(x, norm), w, None, want_stats=True)
        t_sep = timeit(sep)
        ok = ops.conv3x3_gn_fold_ok(x, norm, w)
        t_fold = timeit(lambda: ops.conv3x3_gn_silu(x, norm, w, want_stats=True)) if ok else float("nan")
        err = (ops.conv3x3_gn_silu(x, norm, w) - sep()).abs().max().item() if ok else float("nan")
    print(f"{C}->{co} @{S}^2 x{B}: separate GN pass + conv {t_sep:8.1f} us | folded {t_fold:8.1f} us (gate {'on' if ok else 'off'}) max diff {err:.2e}", flush=True)
