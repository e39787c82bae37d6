#!/usr/bin/env python
"""Micro-benchmark of this repo's HIP kernels at the BASELINE config-2 launch shapes (no UNet): used for
rocprofv3 --pmc passes and A/B timing.  python tools/kbench.py [--rows 8] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stablekeypoints_amd import _native as N, ops  # noqa: E402


def _timed(name, fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / iters * 1e3:9.1f} us")


def conv_part(a, dev, g):
    """Winograd conv3x3 at the heaviest VAE launch shapes of the step and one UNet shape."""
    B = a.rows
    for (ci, co, Hs) in ((128, 128, 512), (512, 512, 128), (1280, 1280, 16)):
        x = torch.randn(B, ci, Hs, Hs, generator=g).to(dev)
        w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(dev)
        U4 = ops._wino4_filters(w, False)
        _timed(f"conv3x3 F(4,3) {ci}->{co} {Hs}^2", lambda: ops._conv3x3_f4_raw(x, U4, None, co), a.iters)
        U2 = ops._wino_filters(w, False)
        _timed(f"conv3x3 F(2,3) {ci}->{co} {Hs}^2", lambda: ops._conv3x3_raw(x, U2, None, co), a.iters)
        del x, w, U4, U2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--what", default="all", choices=["all", "conv", "noconv", "map"])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    if a.what == "conv":
        return conv_part(a, dev, g)
    B, T, R, H = a.rows, a.tokens, a.res, 8
    dims = [(16, 1280)] * 3 + [(32, 640)]
    qs = [torch.randn(B, s * s, C, generator=g).to(dev) for s, C in dims]
    ks = [torch.randn(1, T, C, generator=g).to(dev) for s, C in dims]
    scales = [(C // H) ** -0.5 for _, C in dims]
    S = [ops.qk_logits(q, k, H, sc) for q, k, sc in zip(qs, ks, scales)]
    sides = [s for s, _ in dims]
    M, lse = ops._map_fwd(S, sides, B, H, T, R)
    dM = torch.randn_like(M)
    dS = [torch.empty_like(x) for x in S]
    sp, k1 = N.ptr_array([t.data_ptr() for t in S]); dp, k2 = N.ptr_array([t.data_ptr() for t in dS])
    si, k3 = N.int_array(sides)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(N.lib().skp_attn_map_bwd_workspace(si, 4, B, H, T, R) // 4, device=dev)
    lib = N.lib()

    def timed(name, fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:28s} {e0.elapsed_time(e1) / a.iters * 1e3:9.1f} us")

    timed("attn_map_fwd", lambda: N.check(lib.skp_attn_map_fwd_f32(sp, si, 4, B, H, T, R, M.data_ptr(), lse.data_ptr(), st), "f"))
    timed("attn_map_bwd (A+B)", lambda: N.check(lib.skp_attn_map_bwd_f32(sp, dp, si, 4, B, H, T, R, dM.data_ptr(), lse.data_ptr(), ws.data_ptr(), st), "b"))
    timed("qk_logits x4", lambda: [ops.qk_logits(q, k, H, sc) for q, k, sc in zip(qs, ks, scales)])
    Mt = M[1]
    timed("token_stats", lambda: ops.token_stats(M[0], 1, 2.0))
    am, kl = ops.token_stats(M[0], 1, 2.0); amt, _ = ops.token_stats(Mt, 1, 2.0, want_kl=False)
    timed("select_tokens", lambda: ops.select_tokens(kl, amt[0], R, 25, 10))
    _, sel = ops.select_tokens(kl, amt[0], R, 25, 10)
    timed("fused_losses", lambda: ops.fused_losses(M[0], Mt, sel, am, [0.9, 0.1, 0.05, -0.1, 0.9, -0.02], 2.0, 1))
    for (n, C) in ((4096, 320), (1024, 640)):
        qq, kk, vv, ww = (torch.randn(B, n, C, generator=g).to(dev) for _ in range(4))
        qq.requires_grad_(True); kk.requires_grad_(True); vv.requires_grad_(True)
        timed(f"self_attn_fwd N={n} C={C}", lambda: ops.self_attention(qq, kk, vv, H, (C // H) ** -0.5))
        oo = ops.self_attention(qq, kk, vv, H, (C // H) ** -0.5)
        timed(f"self_attn_bwd N={n} C={C}", lambda: torch.autograd.grad(oo, (qq, kk, vv), ww, retain_graph=True))
    for (n, C) in ((4096, 320), (1024, 640), (256, 1280)):
        q = torch.randn(B, n, C, generator=g).to(dev); k = torch.randn(1, T, C, generator=g).to(dev); v = torch.randn(1, T, C, generator=g).to(dev)
        timed(f"cross_attn_fwd N={n} C={C}", lambda: ops.cross_attention(q, k, v, H, (C // H) ** -0.5))
    if a.what != "noconv":
        conv_part(a, dev, g)


if __name__ == "__main__":
    main()
