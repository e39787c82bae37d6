#!/usr/bin/env python
"""bench.py -- images/sec of the token-optimisation step (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one optimizer step of the reference loop (optimize.py:339-425) on this rank's share of
the global batch: `--images-per-rank` (default 4 = BASELINE config 2's batch_size on 1 GPU) synthetic
512x512 images, each = VAE-encode + hooked UNet forward of the image AND of its affine copy, fused map
reduction x2, on-device token selection, both losses, backward to the [1,77,768] embedding; then one
RCCL all-reduce(sum) of the gradient and an Adam step.
  --scaling weak   (default): per-rank work is fixed, the global batch is images-per-rank * N
  --scaling strong : the global batch is fixed at --global-batch (8 = BASELINE config 3: 1 image per rank at N = 8)
value = global batch * K / max-over-ranks seconds.  `--model sd21|sdxl` run BASELINE configs 4 / 5 (768^2 / 1024^2,
embedding 77 x 1024 / 77 x 2048); `--tokens 500` is the reference CLI default token count.

Rank 0 prints ONE JSON line with `roofline` (the dominant kernel of the step), `roofline_attn_map` /
`roofline_self_attn` (the attention kernels), all measured live with events on the launch stream, `collective_check`
(the RCCL all-reduce proven at start-up) and, at N=1, `cpu_baseline` (the oracle's reference-order CPU step on the
host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

DTYPE_LINE = ("f32 (fp32 MFMA / VALU throughout; flash attention of the >= 1024-key self-attention layers: 3xbf16 operand split, "
              "fp32 accumulate, error <= the fp32 kernel's)")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
F32_MATRIX_PEAK_TF = 157.3     # fp32 MFMA == fp32 vector peak
BF16_MATRIX_PEAK_TF = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md; the 5 PF/s headline is 2:1 sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images-per-rank", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--res", type=int, default=128, help="feature_upsample_res (R)")
    ap.add_argument("--image-size", type=int, default=0, help="0 = the architecture's native size (512 / 768 / 1024)")
    ap.add_argument("--model", default="sd15", choices=["sd15", "sd21", "sdxl", "tiny", "tiny-sd21", "tiny-sdxl"],
                    help="architecture (seeded synthetic weights): sd15 = BASELINE config 2/3, sd21 = config 4, sdxl = config 5")
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--candidates", type=int, default=25, help="furthest_point_num_samples")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --images-per-rank images on every rank (global batch grows with N); strong: the global "
                         "batch is fixed at --global-batch images (BASELINE config 3: 8 images, 1 per rank at N=8)")
    ap.add_argument("--global-batch", type=int, default=8, help="images per optimizer step with --scaling strong")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "on", "off", "full", "sample"],
                    help="auto / on / full: SURVEY 8(d)'s complete protocol (thread sweep, C1 = 256^2 x 50 optimizer steps, "
                         "512^2 x 3 steps; ~6 min of host time); sample: C1 cut to 5 steps (~2 min); off: no CPU leg")
    ap.add_argument("--conv-log", default="", help="write the per-launch list of conv_step_accounting to this JSON file")
    ap.add_argument("--cpu-image-size", type=int, default=512)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = pick the best of a short sweep over {32,64,128} (16, 8 when 32 wins)")
    ap.add_argument("--cache-latents", action="store_true",
                    help="EXPERIMENT (separate line, never the bench of record): latents of the un-warped views kept per dataset image "
                         "(optimize.py cache_latents); the 16-image synthetic set then skips half the VAE work on every timed step")
    ap.add_argument("--kernel-iters", type=int, default=30)
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "file", "off"],
                    help="roofline.traffic: live = two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/traffic_probe.py on "
                         "this box (auto: when rocprofv3 is on PATH and N = 1), file = the committed profiles/r*_pmc_*.json")
    ap.add_argument("--verify", default="auto", choices=["auto", "on", "off"],
                    help="host-drawn weights on BOTH legs and one 256^2 image through the oracle's reference-order CPU step and the "
                         "MI355X step: loss / gradient agreement goes into the line (auto: with the CPU baseline leg)")
    ap.add_argument("--f32-split", "--f32-instr", dest="f32_split", default="auto", choices=["auto", "off"],
                    help="auto: after the timed steps, time the same step with the split-bf16 flash-attention kernels (the "
                         "`f32_split` key; sd* models at N = 1); off: skip it (profiling runs that window on the last steps)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="the step as a captured hipGraph (optimize.GraphedStep): auto = the product's rule (<= 2 images per rank)")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen find)")
    ap.add_argument("--channels-last", action="store_true")
    return ap.parse_args()


def map_kernel_roofline(ops, B, T, R, iters, device, layer_dims=None, top_k=10):
    """Time the fused map kernels alone at the bench shapes (default SD-1.5 hooked layers: 3 x (16^2, C=1280) +
    1 x (32^2, C=640), 8 heads) with events on the CURRENT stream (the one the C-ABI launches on)."""
    g = torch.Generator(device="cpu").manual_seed(0)
    layer_dims = layer_dims or ([(16, 1280, 8)] * 3 + [(32, 640, 8)])
    dims = [(s, C) for s, C, _ in layer_dims]
    qs = [torch.randn(B, s * s, C, generator=g).to(device) for s, C in dims]
    ks = [torch.randn(1, T, C, generator=g).to(device) for s, C in dims]
    H = layer_dims[0][2]
    L = len(dims)
    scales = [(C // H) ** -0.5 for _, C in dims]
    S = [ops.qk_logits(q, k, H, sc) for q, k, sc in zip(qs, ks, scales)]
    sides = [s for s, _ in dims]
    M, lse = ops._map_fwd(S, sides, B, H, T, R)
    # the gradient the step produces: K selected rows per batch row (optimize.py:395-414), dense form for the dense kernels
    Kk = min(top_k, T)
    sel = torch.stack([torch.randperm(T, generator=g)[:Kk] for _ in range(B)]).to(device)
    G = torch.randn(B, Kk, R, R, generator=g).to(device)
    sparse = ops.map_bwd_sparse_supported(sides, Kk, R, T, H)
    dM = torch.zeros_like(M)
    for b in range(B):
        dM[b, sel[b]] = G[b]
    dS = [torch.empty_like(s_) for s_ in S]
    from stablekeypoints_amd import _native as N
    sp, k1 = N.ptr_array([t.data_ptr() for t in S])
    dp, k2 = N.ptr_array([t.data_ptr() for t in dS])
    si, k3 = N.int_array(sides)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(4, N.lib().skp_attn_map_bwd_workspace(si, L, B, H, min(T, 128), R)) // 4, device=device)

    def run_fwd():
        if T > 128:                                              # one-pass wide kernel (or token groups)
            return ops._map_fwd(S, sides, B, H, T, R)
        N.check(N.lib().skp_attn_map_fwd_f32(sp, si, L, B, H, T, R, M.data_ptr(), lse.data_ptr(), st), "fwd")

    def run_bwd():
        if sparse:                                               # the route ops.MapLossesFn takes at this T
            return ops._map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse)
        if T > 128:
            return ops._map_bwd(S, dS, sides, B, H, T, R, dM, lse)
        N.check(N.lib().skp_attn_map_bwd_f32(sp, dp, si, L, B, H, T, R, dM.data_ptr(), lse.data_ptr(), ws.data_ptr(), st), "bwd")

    out = {}
    for name, fn in (("fwd", run_fwd), ("bwd", run_bwd)):
        for _ in range(10):                                      # warm clocks (the GPU idled while the inputs were drawn)
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / iters * 1e-3          # seconds per launch
    # algorithmic bytes per unit: q_low + k + M   (+ dM, dq, dk for backward)
    q_b = sum(s * s * C * 4 for s, C in dims)
    k_b = sum(T * C * 4 for _, C in dims)
    m_b = T * R * R * 4
    fwd_bytes = B * (q_b + m_b) + k_b
    bwd_bytes = B * (m_b + q_b + q_b) + 2 * k_b + k_b
    flops_equiv = B * sum(2 * R * R * C * T for _, C in dims)   # the reference's direct up-res contraction
    out["fwd_kernel"] = ("skp_attn_map_fwd_wide_kernel (one pass, token slices)" if ops.map_wide_supported(T, R) else
                         "skp_attn_map_fwd_kernel" + (" (token groups x two passes)" if T > 128 else ""))
    out["bwd_route"] = (("sparse gradient rows, column sweep (skp_map_bwd_col_kernel: no dV staging)"
                         if T <= ops.COL_MAX_T and ops.map_bwd_col_supported(sides, Kk, R, T, H) else
                         "sparse gradient rows, token-major sweep (skp_map_bwd_tok_kernel)") if sparse else
                        "dense gradient (skp_attn_map_bwd_kernel + vadj)")
    return out, fwd_bytes, bwd_bytes, flops_equiv


def self_attn_roofline(ops, B, iters, device):
    """Flash self-attention at the 64^2 layers of the step (N=4096 tokens, 8 heads x d=40, B rows): algorithmic
    FLOPs = 4*N^2*d per (row, head) forward (QK^T + PV), 2.5x that backward (5 products), against the fp32 MFMA peak."""
    g = torch.Generator(device="cpu").manual_seed(1)
    Nq, H, d = 4096, 8, 40
    q, k, v, w = (torch.randn(B, Nq, H * d, generator=g).to(device) for _ in range(4))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    out = {}
    fwd = lambda: ops.self_attention(q, k, v, H, d ** -0.5)
    for _ in range(8):                                           # the GPU idles while the inputs are drawn on the host: let the
        o = fwd()                                                # clocks come back up before timing (5 calls read 10 % slow)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        o = fwd()
    e1.record(); torch.cuda.synchronize()
    out["fwd"] = e0.elapsed_time(e1) / iters * 1e-3
    o = fwd()
    for _ in range(4):
        torch.autograd.grad(o, (q, k, v), w, retain_graph=True)
    e0.record()
    for _ in range(iters):
        torch.autograd.grad(o, (q, k, v), w, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    out["bwd"] = e0.elapsed_time(e1) / iters * 1e-3
    flops_fwd = 4.0 * Nq * Nq * d * B * H
    return out, flops_fwd, 2.5 * flops_fwd


def conv_roofline(ops, B, image_size, iters, device):
    """The Winograd F(4x4,3x3) conv kernel at the heaviest launch shape of the step, IN THE FORM THE STEP LAUNCHES IT: the VAE
    encoder's first-level ResnetBlock2D convs (128 -> 128 channels at image resolution, B rows) run as
    skp_wino4_conv_c128_kernel<STATS = true, GNF = true> -- GroupNorm + SiLU of the input applied in the patch load, block
    statistics for the next GroupNorm left behind by the epilogue (launches #0-#3 of a step).  Executed MFMA FLOPs =
    direct-form FLOPs / 4 (36 multiplies per 4x4 output tile and channel pair instead of 144); algorithmic bytes = input +
    output + filter.  Where the folded form does not serve the shape (reduced-width test models) the plain form is timed."""
    lib, N = ops.N.lib(), ops.N
    g = torch.Generator(device="cpu").manual_seed(2)
    ci = co = 128
    B = min(B, max(1, (2 ** 31 - 1) // (ci * image_size * image_size * 4)))   # one launch (the op chunks larger batches)
    x = torch.randn(B, ci, image_size, image_size, generator=g).to(device)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(device)
    U = ops._wino4_filters(w, False)
    nblk = ops.conv3x3_stats_blocks(x.shape, w.shape)
    folded = bool(nblk) and bool(lib.skp_conv3x3_f4_gn_ok(B, ci, co, image_size, image_size))
    if folded:
        y = torch.empty(B, co, image_size, image_size, device=device)
        stats = torch.empty(B, co, nblk, 2, device=device)
        coef = torch.stack([torch.full((B, ci), 0.7), torch.full((B, ci), 0.1)], dim=-1).to(device).contiguous()
        fn = lambda: N.check(lib.skp_conv3x3_f4_gn_f32(x.data_ptr(), U.data_ptr(), None, None, y.data_ptr(), stats.data_ptr(),
                                                       coef.data_ptr(), B, ci, co, image_size, image_size, ops._stream()),
                             "skp_conv3x3_f4_gn_f32")
    else:
        fn = lambda: ops._conv3x3_f4_raw(x, U, None, co)
    for _ in range(8):                                           # as above: warm clocks before timing
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    direct = 2.0 * 9 * ci * co * B * image_size * image_size
    nbytes = 4 * (B * ci * image_size ** 2 + B * co * image_size ** 2 + 36 * ci * co)
    units = 8 * ((((B * (image_size // 4) ** 2 + 15) // 16) + 7) // 8) * (co // 128)
    grid_threads = min(units, 256) * 256                         # 128-channel form: persistent, one workgroup per CU
    return t, direct, nbytes, grid_threads, B, folded


def split_flash_probe(ops, B, iters, device):
    """The flash-attention forward of the 64^2 layers (N = 4096, 8 heads x 40) on the bf16 matrix cores with three-term operand
    splits (csrc/skp_flash_attn_s.hip, its K / V pre-pass included) next to the fp32-instruction forward, and both errors against
    an fp64 reference on one (row, head)."""
    g = torch.Generator(device="cpu").manual_seed(4)
    H, N, d = 8, 4096, 40
    q, k, v = (torch.randn(B, N, H * d, generator=g).to(device) for _ in range(3))
    scale = d ** -0.5
    o32 = torch.empty_like(q); l32 = torch.empty(B, H, N, device=device)
    lib, Nn = ops.N.lib(), ops.N
    f32 = lambda: Nn.check(lib.skp_flash_attn_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), o32.data_ptr(), l32.data_ptr(), B, B, H, N, N, d,
                                                      float(scale), ops._stream()), "fp32 flash forward")
    spl = lambda: ops.flash_attn_fwd_split(q, k, v, H, scale)
    for _ in range(3):
        f32(); spl()
    times = {"f32": [], "split": []}
    for _ in range(3):
        for name, fn in (("f32", f32), ("split", spl)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / iters * 1e3)
    ref = ((q[0, :, :d].double() @ k[0, :, :d].double().T) * scale).softmax(-1) @ v[0, :, :d].double()
    f32(); os_, _ = spl()
    e32 = (o32[0, :, :d].double() - ref).abs().max().item(); esp = (os_[0, :, :d].double() - ref).abs().max().item()
    t32, tsp = sorted(times["f32"])[1], sorted(times["split"])[1]
    return {"what": f"flash attention forward, N = {N}, {H} heads x {d}, {B} rows: fp32-instruction kernel vs the split-bf16 kernel (pre-pass included)",
            "fp32_kernel_us": t32, "split_kernel_us": tsp, "speedup": t32 / tsp, "max_err_vs_fp64_ratio": esp / e32,
            "max_err_vs_fp64": {"fp32_kernel": e32, "split_kernel": esp},
            "split_tflops_equiv": 4.0 * N * N * d * B * H / tsp / 1e6}


def conv_step_forms(ops, B, image_size, iters, device):
    """The same kernel in the forms and at the launch shapes the step actually runs (micro-timed like conv_roofline): the
    first-level VAE convolutions carry the GroupNorm + SiLU of their input in the patch load and leave block statistics
    for the next GroupNorm behind; the deeper levels are the plain kernel with the statistics epilogue."""
    lib, N = ops.N.lib(), ops.N
    g = torch.Generator(device="cpu").manual_seed(3)
    out = []
    for (ci, co, sz, form) in ((128, 128, image_size, "plain"), (128, 128, image_size, "gn_fold+stats"),
                               (256, 256, image_size // 2, "stats"), (512, 512, image_size // 4, "stats"),
                               (320, 320, image_size // 8, "stats")):
        rows = min(B, max(1, (2 ** 31 - 1) // (max(ci, co) * sz * sz * 4)))
        x = torch.randn(rows, ci, sz, sz, generator=g).to(device)
        w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(device)
        U = ops._wino4_filters(w, False)
        nblk = ops.conv3x3_stats_blocks(x.shape, w.shape)
        if not nblk:
            continue
        stats = torch.empty(rows, co, nblk, 2, device=device)
        y = torch.empty(rows, co, sz, sz, device=device)
        if form.startswith("gn_fold"):
            if not lib.skp_conv3x3_f4_gn_ok(rows, ci, co, sz, sz):
                continue
            coef = torch.stack([torch.full((rows, ci), 0.7), torch.full((rows, ci), 0.1)], dim=-1).to(device).contiguous()
            fn = lambda: N.check(lib.skp_conv3x3_f4_gn_f32(x.data_ptr(), U.data_ptr(), None, None, y.data_ptr(), stats.data_ptr(),
                                                           coef.data_ptr(), rows, ci, co, sz, sz, ops._stream()), "skp_conv3x3_f4_gn_f32")
        elif form == "plain":                                     # rounds 1-3 headline form: no statistics, no folded norm
            fn = lambda: ops._conv3x3_f4_raw(x, U, None, co, out=y)
        else:
            fn = lambda: ops._conv3x3_f4_raw(x, U, None, co, out=y, stats=stats)
        for _ in range(6):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e-3
        fl = 2.0 * 9 * ci * co * rows * sz * sz / 4
        out.append({"shape": f"{ci}->{co} ch at {sz}^2, {rows} rows", "form": form, "launch_us": t * 1e6, "algorithmic_flops": fl,
                    "achieved": fl / t / 1e12, "frac": fl / t / 1e12 / F32_MATRIX_PEAK_TF})
        del x, y, stats, U, w
    return out


def conv_step_accounting(ops, one_step, log_path=None):
    """Every Winograd F(4x4,3x3) launch of ONE optimizer step (VAE encoder + UNet forward + UNet backward-data), timed with an
    event pair on the launch stream around each C-ABI call (the K-split reduction of a split launch and the input-transform
    kernel of the raw-filter form included), with its
    algorithmic FLOPs = 2*9*Cin*Cout*B*H*W / 4: the TIME-WEIGHTED fraction of the fp32 matrix peak over all of them
    (sum of FLOPs / sum of time / 157.3) is the number the per-shape `roofline.frac` cannot give.  The launch list goes to
    `log_path` (tools/step_breakdown.py joins it with a rocprofv3 kernel trace by launch order)."""
    lib = ops.N.lib()
    names = {"skp_conv3x3_f4_f32": 6, "skp_conv3x3_f4_stats_f32": 6, "skp_conv3x3_f4_gn_f32": 7,
             "skp_conv3x3_f4r_f32": 6}                                                          # index of B in the arguments
    real = {n: getattr(lib, n) for n in names}
    log = []

    def wrap(name, fn, ib):
        def call(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            log.append((name, tuple(int(v) for v in args[ib:ib + 5]), e0, e1))
            return rc
        return call

    one_step()                                                   # steady state first (wrappers off)
    try:
        for n, ib in names.items():
            setattr(lib, n, wrap(n, real[n], ib))
        one_step()
    finally:
        for n in names:
            setattr(lib, n, real[n])
    torch.cuda.synchronize()
    rows, by_shape = [], {}
    for name, (B, ci, co, H, W), e0, e1 in log:
        us = e0.elapsed_time(e1) * 1e3
        fl = 2.0 * 9 * ci * co * B * H * W / 4
        rows.append({"entry": name, "B": B, "Cin": ci, "Cout": co, "H": H, "W": W, "algorithmic_flops": fl, "us": us})
        form = {"skp_conv3x3_f4_f32": "plain", "skp_conv3x3_f4_stats_f32": "stats", "skp_conv3x3_f4_gn_f32": "gn",
                "skp_conv3x3_f4r_f32": "raw-filter"}[name]
        d = by_shape.setdefault((form, ci, co, H, W, B), [0, 0.0, 0.0])
        d[0] += 1; d[1] += fl; d[2] += us
    if not rows:
        return None
    tf, tus = sum(r["algorithmic_flops"] for r in rows), sum(r["us"] for r in rows)
    top = sorted(by_shape.items(), key=lambda kv: -kv[1][2])[:12]
    out = {"what": "all Winograd F(4x4,3x3) launches of one optimizer step (forward + backward-data, K-split reductions "
                   "included), one event pair per C-ABI call on the launch stream",
           "launches": len(rows), "algorithmic_flops": tf, "ms": tus / 1e3, "achieved": tf / tus / 1e6, "peak": F32_MATRIX_PEAK_TF,
           "unit": "TFLOP/s", "frac": tf / tus / 1e6 / F32_MATRIX_PEAK_TF,
           "heaviest_shapes": [{"entry": k[0], "shape": f"{k[1]}->{k[2]} ch at {k[3]}x{k[4]}, {k[5]} rows", "calls": v[0],
                                "ms": v[2] / 1e3, "frac": v[1] / v[2] / 1e6 / F32_MATRIX_PEAK_TF} for k, v in top]}
    if log_path:
        with open(log_path, "w") as f:
            json.dump({"launches": rows, "summary": {k: out[k] for k in ("launches", "algorithmic_flops", "ms", "achieved", "frac")}}, f)
    return out


def cpu_baseline(ldm_cpu, args):
    """Oracle reference-order CPU step (oracle/cpu_path.py: materialised attention, x-upsample + second to_q, stack+mean
    collect_maps, python selection, torch losses, Adam) on the host cores -- a BOUNDED sample:
      default: thread sweep {32,64,128} (and 16, 8 when 32 wins) on one 256^2 image each, then at the best count: C1 shape (256^2, batch 1) for
               the protocol's full 50 optimizer steps and the bench shape (512^2) for 1 warm-up + 3 timed steps;
      --cpu-baseline sample: C1 cut to 5 steps (labelled `sampled`).
    `value` is the bench-shape rate (same image size as the GPU line); the C1 rate rides along."""
    from oracle import cpu_path
    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))                    # the cores this process may actually run on (cgroup / taskset)
    except AttributeError:
        usable = ncpu
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(1, args.tokens, ldm_cpu.unet.config["cross_attention_dim"], generator=g)
    kw = dict(R_up=args.res, furthest_point_num_samples=args.candidates, top_k=args.top_k)

    def run(size, steps, threads):
        torch.set_num_threads(threads)
        imgs = torch.rand(max(1, steps), 3, size, size, generator=torch.Generator().manual_seed(1))
        _, sec, n = cpu_path.optimize_embedding_cpu(ldm_cpu, imgs, ctx, steps=steps, batch_size=1, **kw)
        return n / sec, sec

    sweep = {}
    if args.cpu_threads > 0:
        best = min(ncpu, args.cpu_threads)
    else:
        run(128, 1, min(ncpu, 32))                               # allocator / first-touch warm-up, untimed
        for th in (32, 64, 128):
            if th <= ncpu:
                sweep[th] = run(256, 1, th)[0]
                if len(sweep) > 1 and sweep[th] < 0.8 * max(sweep.values()):
                    break                                        # past the knee: more threads only oversubscribe
        if sweep and max(sweep, key=sweep.get) == min(sweep):    # best at the low end of the sweep: look below it too
            for th in (16, 8):
                if th < min(sweep):
                    sweep[th] = run(256, 1, th)[0]
                    if sweep[th] < max(sweep.values()):
                        break
        best = max(sweep, key=sweep.get) if sweep else min(ncpu, 32)
    c1_steps = 5 if args.cpu_baseline == "sample" else 50
    c1_rate, c1_sec = run(256, c1_steps, best)
    size = args.cpu_image_size
    run(size, 1, best)                                           # warm-up at the bench shape
    rate, sec = run(size, 3, best)
    return {"value": rate, "unit": "images/sec", "cores": best, "kind": "port",
            "sample": f"{size}x{size}: 1 warm-up + 3 timed optimizer steps (batch 1: 2 VAE+UNet forwards with materialised "
                      f"attention, backward, Adam per step), T={args.tokens}, R={args.res}, {sec:.1f} s; torch "
                      f"{torch.__version__} CPU fp32, {best} threads of {ncpu} logical CPUs ({usable} in this process's affinity mask); "
                      "the thread count is the winner of the sweep below, not the host's size: past it the op-level parallel "
                      "regions of the reference-order path oversubscribe and slow down",
            "host_cpus": {"os_cpu_count": ncpu, "sched_getaffinity": usable},
            "c1_256": {"value": c1_rate, "steps": c1_steps, "seconds": c1_sec, "sampled": c1_steps < 50,
                       "full_protocol_steps": 50,
                       "what": "BASELINE config 1 shape (256^2, 1 image, batch 1), reference op order"
                               + ("" if c1_steps >= 50 else f" -- a {c1_steps}-step SAMPLE of the 50-step protocol "
                                  "(the default runs all 50)")},
            "thread_sweep_256_images_per_sec": {str(k): v for k, v in sweep.items()}}


VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9     # fp32 VALU issue: 256 CUs x 4 SIMDs x 16 lanes per clock at 2.4 GHz (a wave instruction = 4 clk;
                                                # packed fp32 is half rate on this part, profiles/r02_mfma_valu_probe.md: no second factor)


def map_valu_roof(B, L, H, T, R, fwd_s):
    """The roof that BINDS the fused map forward: VALU issue.  Per (layer, head, token, up-res pixel) element the kernel
    (csrc/skp_attn_map.hip, skp_attn_map_fwd_kernel, quad layers) issues, in lane-clocks: horizontal bicubic 1 mul + 3 DPP fmacs
    = 4; running max 1; subtract the max 1 (packed: two tokens per instruction at half rate); exp2 4 (quarter-rate
    transcendental); row sum 1; normalise + accumulate 1 = 12.  Pad tokens (NT = 16 ceil(T / 16)) are computed too.  The V phase
    (vertical taps into LDS, shared by the 4..8 pixels under a low-res row) and the stores add ~5 %, not modelled."""
    nt = (T + 15) // 16 * 16
    elements = float(B) * L * H * nt * R * R
    ops = {"horizontal bicubic (mul + 3 DPP fmac)": 4, "running max": 1, "subtract max": 1, "exp2 (quarter rate)": 4, "row sum": 1,
           "normalise + accumulate": 1}
    per = sum(ops.values())
    floor = elements * per / VALU_LANE_OPS_PER_S
    counted = None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_map*.json")), reverse=True):
        try:
            for name, c in json.load(open(path))["kernels"].items():
                if "skp_attn_map_fwd_kernel" in name and "SQ_INSTS_VALU" in c:
                    counted = {"source": "profiles/" + os.path.basename(path), "SQ_INSTS_VALU_per_launch": c["SQ_INSTS_VALU"],
                               "issue_floor_us": c["SQ_INSTS_VALU"] * 4 / (256 * 4) / 2.4e9 * 1e6,
                               **{k: c[k] for k in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES") if k in c}}
                    break
        except Exception:
            continue
        if counted:
            break
    return {"bound": "valu", "elements": elements, "lane_clocks_per_element": per, "model": ops, "peak_lane_ops_per_s": VALU_LANE_OPS_PER_S,
            "floor_us": floor * 1e6, "launch_us": fwd_s * 1e6, "frac": floor / fwd_s, "counters": counted}


def measured_traffic(kernel_substr, template=None):
    """(HBM bytes per launch, source file) of a kernel from the committed rocprofv3 --pmc passes (profiles/*.json: separate
    FETCH_SIZE / WRITE_SIZE runs of tools/kbench.py at the same launch shape, FETCH_SIZE doubled per
    MI355X_MICROARCH.md 'HBM'; regenerate with tools/pmc_traffic.sh).  (None, None) when no profile is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), reverse=True)       # newest round / version first by name
    for path in files:
        try:
            k = json.load(open(path))["kernels"]
        except Exception:
            continue
        for name, c in k.items():
            plain = re.sub(r"<[^<>]*>", "", name).replace("void ", "")          # template arguments / return type of the trace name
            if template is not None and f"<{template}>" not in name.replace(" ", ""):      # the persistent conv kernel: one grid, several forms
                continue
            if kernel_substr in plain and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "profiles/" + os.path.basename(path)
    return None, None


def live_traffic(a, rows, image_size):
    """HBM bytes per launch of the roofline kernels measured NOW: two counter-only rocprofv3 passes (FETCH_SIZE, then
    WRITE_SIZE -- they do not fit one pass; never combined with a trace) over tools/traffic_probe.py.  FETCH_SIZE is
    doubled per MI355X_MICROARCH.md (HBM section).  -> {kernel name: bytes} or None."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    probe = [sys.executable, os.path.join(ROOT, "tools", "traffic_probe.py"), "--rows", str(rows), "--tokens", str(a.tokens),
             "--res", str(a.res), "--image-size", str(min(image_size, 512)), "--top-k", str(a.top_k)]
    acc = {}
    tmp = tempfile.mkdtemp(prefix="skp_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "-f", "csv", "-d", out, "-o", "p", "--"] + probe, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            per = {}
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
                    if "skp_" not in k:
                        continue
                    if "_conv_" in k:
                        k += "@grid" + row["Grid_Size"]
                    d = per.setdefault(k, {})
                    d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            for k, d in per.items():
                acc.setdefault(k, {})[counter] = sum(d.values()) / len(d)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {k: int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024) for k, c in acc.items()
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c}


def verify_against_oracle(ldm_cpu, ldm, controller, a, dev):
    """One 256^2 image (BASELINE configs[0]'s shape) through the oracle's reference-order CPU step and through the MI355X
    step, SAME weights (drawn on the host, copied to the GPU), same noise and affine: the line carries the agreement."""
    from oracle import cpu_path, ref_path as R
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import default_args, group_step
    g = torch.Generator().manual_seed(5)
    size = 256 if a.model.startswith("sd") else 128
    image = torch.rand(1, 3, size, size, generator=g)
    width = ldm_cpu.unet.config["cross_attention_dim"]
    ctx = torch.randn(1, a.tokens, width, generator=g) * 5.0
    noise = torch.randn(2, 4, size // 8, size // 8, generator=g)
    theta = R.affine_matrix(9.0, 0.9, (0.1, -0.15))
    args = default_args(num_tokens=a.tokens, feature_upsample_res=a.res, furthest_point_num_samples=a.candidates,
                        top_k=a.top_k, batch_size=1)
    store = R.OracleStore()
    cpu_path.register_reference_hook(ldm_cpu.unet, store, a.res)
    c_ref = ctx.clone().requires_grad_(True)
    t0 = time.perf_counter()
    loss, sharp, equiv, sel, am, am_t = cpu_path.image_step(ldm_cpu, image, c_ref, store, theta, noise[0:1], noise[1:2],
                                                            furthest_point_num_samples=a.candidates, top_k=a.top_k,
                                                            sigma=args.sigma)
    loss.backward()
    cpu_s = time.perf_counter() - t0
    c_gpu = ctx.clone().to(dev).requires_grad_(True)
    lg, eg, sg = group_step(ldm, image, c_gpu, args, controller, RandomAffineWithInverse(), denom=1, noise=noise.to(dev),
                            thetas=theta)
    gref, ggpu = c_ref.grad, c_gpu.grad.cpu()
    loss, sharp, equiv, lg, sg, eg = (torch.as_tensor(v).detach() for v in (loss, sharp, equiv, lg, sg, eg))
    return {"what": f"1 image {size}x{size}, T={a.tokens}, R={a.res}: oracle reference-order CPU step vs the MI355X step, same "
                    "host-drawn weights / noise / affine",
            "loss_cpu": float(loss), "loss_gpu": float(lg), "loss_rel_diff": abs(float(lg) - float(loss)) / abs(float(loss)),
            "sharpening_rel_diff": abs(float(sg) - float(sharp)) / abs(float(sharp)),
            "equivariance_rel_diff": abs(float(eg) - float(equiv)) / abs(float(equiv)),
            "grad_max_abs_diff_over_max": float((ggpu - gref).abs().max() / gref.abs().max()),
            "grad_cosine": float(torch.nn.functional.cosine_similarity(ggpu.flatten(), gref.flatten(), dim=0)),
            "cpu_step_seconds": cpu_s}


NATIVE_SIZE = {"sd15": 512, "sd21": 768, "sdxl": 1024, "tiny": 128, "tiny-sd21": 128, "tiny-sdxl": 128}
MODEL_LABEL = {"sd15": "SD1.5", "sd21": "SD2.1", "sdxl": "SDXL"}
CONFIG_NAME = {"sd15": "BASELINE config 2", "sd21": "BASELINE config 4", "sdxl": "BASELINE config 5"}


def hooked_layer_dims(model, image_size):
    """(side, channels, heads) of the layers the <= 32^2 / first-4 gate stores."""
    lat = image_size // 8
    if model == "sd21":
        return [(lat // 4, 1280, 20)] * 3
    if model == "sdxl":
        return [(lat // 4, 1280, 20)] * 4
    if model == "sd15":
        return [(lat // 4, 1280, 8)] * 3 + [(lat // 2, 640, 8)]
    return None


def rccl_self_check(D, world, rank, dev):
    """Prove the collective this run depends on: all-reduce(SUM) of a one-hot rank vector must come back as all ones,
    and of the rank ids as 0+1+...+N-1.  Returns what was checked (goes into the JSON line)."""
    if world == 1:
        return {"backend": None, "ok": True, "ranks_seen": 1}
    v = torch.zeros(world + 1, device=dev, dtype=torch.float32)
    v[rank] = 1.0
    v[world] = float(rank)
    D.allreduce_sum_(v)
    torch.cuda.synchronize()
    ok = bool((v[:world] == 1).all().item()) and float(v[world].item()) == world * (world - 1) / 2
    assert ok, f"RCCL self-check failed on rank {rank}: {v.tolist()}"
    return {"backend": torch.distributed.get_backend(), "ok": True, "ranks_seen": int(v[:world].sum().item())}


def self_launch_command(gpus, argv, port=None):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: the command that re-runs this file under
    `torch.distributed.run`, one rank per GPU on 127.0.0.1 (the driver's own N > 1 command line, verbatim)."""
    if port is None:
        import socket
        with socket.socket() as s:                               # a port the kernel says is free right now
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torchrun around us: become the launcher (exit status = torchrun's: non-zero if any rank dies; rank 0 of
        # the relaunched job prints the one JSON line on the inherited stdout)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC is the only form the host driver supports
        cmd = self_launch_command(a.gpus, sys.argv[1:])
        print("bench.py: --gpus %d without a launcher environment -> %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
    from stablekeypoints_amd import dist as D
    world, rank, local = D.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    if os.environ.get("SKP_BENCH_SINGLE_DEVICE") == "1":       # test hook: several ranks share cuda:0 (with gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # host threads: N ranks share the box's cores (the Python driver is the only CPU work of a rank)
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 8) // max(1, world))))
    from stablekeypoints_amd import ops, _native
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import SyntheticImages, default_args, group_step
    from stablekeypoints_amd.optimize_token import load_ldm
    _native.lib()
    coll = rccl_self_check(D, world, rank, dev)
    image_size = a.image_size or NATIVE_SIZE[a.model]
    if a.cpu_image_size == 512 and image_size != 512:
        a.cpu_image_size = image_size

    t_build = time.time()
    cpu_stats, t_cpu = None, 0.0
    want_cpu = a.cpu_baseline in ("on", "full", "sample") or (a.cpu_baseline == "auto" and world == 1 and a.model == "sd15")
    want_verify = a.verify == "on" or (a.verify == "auto" and want_cpu)
    ldm_cpu = None
    if (want_cpu or want_verify) and rank == 0:
        # the CPU leg gets its own host-side instance (seeded host draw)
        ldm_cpu, _, _ = load_ldm("cpu", a.model, feature_upsample_res=a.res)
        if want_cpu:
            t_cpu = time.time()
            cpu_stats = cpu_baseline(ldm_cpu, a)
            t_cpu = time.time() - t_cpu
    # frozen weights are drawn on this rank's GPU (seeded: identical on every rank; no N x 3.4 GB of host-side init)
    ldm, controllers, _ = load_ldm(dev, a.model, feature_upsample_res=a.res, init_on_device=True)
    controller = controllers[dev]
    verify = None
    weights_init = "seeded, drawn on the rank's GPU"
    if want_verify and rank == 0 and world == 1:
        with torch.no_grad():                                    # BOTH legs on the host-drawn weights
            for src, dst in ((ldm_cpu.unet, ldm.unet), (ldm_cpu.vae, ldm.vae)):
                sd = src.state_dict()
                for k, v in dst.state_dict().items():
                    v.copy_(sd[k])
        weights_init = "seeded, drawn on the host, copied to the GPU (the CPU legs use the same tensors)"
        verify = verify_against_oracle(ldm_cpu, ldm, controller, a, dev)
    del ldm_cpu
    if a.miopen_find:
        torch.backends.cudnn.benchmark = True
    if a.channels_last:
        ldm.unet.to(memory_format=torch.channels_last)
        ldm.vae.to(memory_format=torch.channels_last)
    from stablekeypoints_amd import tuning
    gemm_tuned = tuning.enable()
    t_build = time.time() - t_build - t_cpu            # model construction, weight draw, verify leg (not the CPU baseline)

    if a.scaling == "strong":
        if a.global_batch % world:
            raise SystemExit(f"--scaling strong: --global-batch {a.global_batch} must be a multiple of --gpus {world}")
        global_batch, per_rank = a.global_batch, a.global_batch // world
    else:
        per_rank = a.images_per_rank
        global_batch = per_rank * world
    width = ldm.unet.config["cross_attention_dim"]
    args = default_args(num_tokens=a.tokens, feature_upsample_res=a.res, batch_size=global_batch, device=str(dev),
                        image_size=image_size, top_k=a.top_k, furthest_point_num_samples=a.candidates)
    data = SyntheticImages(n=max(16, per_rank * 2), size=image_size, seed=rank, device=dev)
    torch.manual_seed(1000 + rank)                                                    # per-rank augmentations/noise
    ctx = torch.randn(1, a.tokens, width, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(True)
    opt = torch.optim.Adam([ctx], lr=args.lr)
    reducer = D.EmbeddingReducer(ctx, opt)
    transform = RandomAffineWithInverse(args.augment_degrees, args.augment_scale, args.augment_translate)

    cursor = 0
    latent_cache = {} if a.cache_latents else None

    # small per-rank batches are launch-bound on the host: the product captures their step in a hipGraph (optimize.GraphedStep,
    # `capture_step="auto"` = groups of <= 2 images); the bench follows the product's rule unless told otherwise
    from stablekeypoints_amd.optimize import GraphedStep
    use_graph = latent_cache is None and (a.graph == "on" or (a.graph == "auto" and per_rank <= 2))
    graphed = GraphedStep(ldm, ctx, args, controller, transform, global_batch, warn_route=(a.graph == "on")) if use_graph else None

    def one_step(eager=False):
        nonlocal cursor
        idx = [(cursor + i) % len(data) for i in range(per_rank)]
        cursor += per_rank
        images = torch.stack([data[i]["img"] for i in idx])
        if graphed is not None and not eager:
            out = graphed(images)
        else:
            out = group_step(ldm, images, ctx, args, controller, transform, denom=global_batch, latent_cache=latent_cache, ids=idx)
        reducer.step()
        return out

    one_step()                       # untimed pre-warm: MIOpen/hipBLASLt first-call solver selection, allocator growth
    for _ in range(a.warmup + (len(data) // per_rank if a.cache_latents else 0)):     # experiment: first epoch fills the cache
        one_step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c0 = time.thread_time()                                       # CPU time of the launch thread (the only host work of a rank)
    for _ in range(a.steps):
        last = one_step()
    c_launch = time.thread_time() - c0                            # ... until its last launch is queued (the waits below are idle time)
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    # one more step from an idle GPU: when does the launch thread return, when is the GPU done?  (launch_ms < total_ms: the
    # step is GPU-bound and the host has slack; launch_ms ~ total_ms: launch-bound -- what a captured hipGraph would remove)
    ti = time.perf_counter()
    one_step()
    t_issue = time.perf_counter() - ti
    torch.cuda.synchronize()
    t_idle_total = time.perf_counter() - ti
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(tmax.item())
    # embeddings must still be bit-identical on every rank (same all-reduced gradient, same Adam state)
    chk = torch.stack([ctx.detach().double().sum(), ctx.detach().double().abs().sum()])
    if world > 1:
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        coll["embedding_identical_on_all_ranks"] = bool(torch.equal(lo, hi))

    if rank == 0:
        B = 2 * per_rank                                          # rows per fused-map launch (both views)
        ldims = hooked_layer_dims(a.model, image_size)
        kt, fwd_bytes, bwd_bytes, flops_equiv = map_kernel_roofline(ops, B, a.tokens, a.res, a.kernel_iters, dev, ldims, a.top_k)
        sa, sa_f, sa_b = self_attn_roofline(ops, B, max(10, a.kernel_iters // 3), dev)
        cv_t, cv_direct, cv_bytes, cv_grid, cv_rows, cv_folded = conv_roofline(ops, B, min(image_size, 512), max(10, a.kernel_iters // 3), dev)
        cv_forms = conv_step_forms(ops, B, min(image_size, 512), max(10, a.kernel_iters // 3), dev)
        f32_instr = None
        if a.model.startswith("sd") and world == 1 and a.f32_split != "off":
            # Beside the line of record (never part of `value`): the same step with the flash attention of the 64^2 / 32^2
            # self-attention layers on the fp32-INSTRUCTION kernels (rounds 1-5's route; the line runs them on the bf16 matrix cores
            # with three-term operand splits, fp32 accumulate), and the two forward kernels side by side with their errors vs fp64.
            try:
                ops.FLASH_SPLIT = False
                for _ in range(2):
                    one_step(eager=True)
                torch.cuda.synchronize()
                t0s = time.perf_counter()
                for _ in range(a.steps):
                    one_step(eager=True)
                torch.cuda.synchronize()
                el_s = time.perf_counter() - t0s
                ops.FLASH_SPLIT = True
                fa_probe = split_flash_probe(ops, B, max(5, a.kernel_iters // 6), dev)
                f32_instr = {"value": global_batch * a.steps / el_s, "ms_per_step": el_s / a.steps * 1e3, "unit": "images/sec",
                             "dtype": "f32 (v_mfma_f32_16x16x4_f32 in every matrix kernel)",
                             "what": "the same step with ops.FLASH_SPLIT = False: self-attention of the 64^2 / 32^2 layers on the "
                                     "fp32-instruction flash kernels (forward at d = 40 / 80, backward at d = 40 are the launches that differ)",
                             "split_max_err_vs_fp64_over_fp32_kernel": fa_probe["max_err_vs_fp64_ratio"],
                             "flash_forward": fa_probe}
            except Exception as e:                                # noqa: BLE001 -- a comparison key never takes the line of record down
                print(f"bench.py: f32_instr comparison unavailable ({e})", file=sys.stderr)
                f32_instr = None
            finally:
                ops.FLASH_SPLIT = True
        def local_step():                                        # rank 0 only: the step without its collective
            nonlocal cursor
            idx = [(cursor + i) % len(data) for i in range(per_rank)]
            cursor += per_rank
            group_step(ldm, torch.stack([data[i]["img"] for i in idx]), ctx, args, controller, transform, denom=global_batch,
                       latent_cache=latent_cache, ids=idx)
            ctx.grad = None
        conv_all = conv_step_accounting(ops, local_step, a.conv_log or None)
        ach = fwd_bytes / kt["fwd"] / 1e9
        value = global_batch * a.steps / elapsed
        try:                                                      # measurement aid from tools/csrc/libskp_lab.so (not the product library)
            issue1, issue2 = ops.mfma_issue_rate(1, device=dev), ops.mfma_issue_rate(2, device=dev)
        except Exception as e:                                    # noqa: BLE001 -- the line of record does not depend on the probe
            print(f"bench.py: MFMA issue-rate probe unavailable ({e})", file=sys.stderr)
            issue1 = issue2 = None
        conv_traffic, conv_src = measured_traffic(f"skp_wino4_conv_c128_kernel@grid{cv_grid}",
                                                  template="true,true" if cv_folded else "false,false")
        if conv_traffic is None:                                 # committed passes from before the kernel became persistent: same
            conv_traffic, conv_src = measured_traffic("skp_wino4_conv_c128_kernel@grid2097152")    # launch shape, one workgroup per unit
        map_kernel = "skp_attn_map_fwd_wide_kernel" if ops.map_wide_supported(a.tokens, a.res) else "skp_attn_map_fwd_kernel"
        map_traffic, map_src = measured_traffic(map_kernel, template=None if "wide" in map_kernel else f"{(a.tokens + 15) // 16 * 16},0")
        live, map_bwd_traffic = None, None
        if a.traffic == "live" or (a.traffic == "auto" and world == 1 and a.model == "sd15"):
            torch.cuda.empty_cache()
            live = live_traffic(a, B, image_size)
        if live:
            src = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/traffic_probe.py in this run (2 x FETCH + WRITE)"
            for k, v in live.items():
                if "skp_wino4_conv_c128_kernel" in k:
                    conv_traffic, conv_src = v, src
                if map_kernel in k:
                    map_traffic, map_src = v, src
            bwd_parts = {k: v for k, v in live.items() if any(t in k for t in ("map_bwd", "map_dot", "map_vadj", "map_taps"))}
            if bwd_parts:                                        # every kernel of the backward route launches once per backward
                map_bwd_traffic = {"total": sum(bwd_parts.values()), "by_kernel": bwd_parts}
        elif a.traffic == "off":
            conv_traffic = conv_src = map_traffic = map_src = None
        cfg_name = CONFIG_NAME.get(a.model, "reduced-width test model")
        if a.model == "sd15" and a.scaling == "strong":
            cfg_name = "BASELINE config 3 (fixed global batch)"
        line = {
            "metric": f"images/sec for token-optimization step ({MODEL_LABEL.get(a.model, a.model)}, {image_size}^2, K={a.top_k} kpts)",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "scaling_definition": ("weak: every rank processes images_per_rank images per optimizer step, the global batch "
                                   "is images_per_rank x N" if a.scaling == "weak" else
                                   "strong: the global batch is fixed (global_batch images per optimizer step), each rank "
                                   "processes global_batch / N of them"),
            "vs_baseline": None,
            "dtype": DTYPE_LINE,
            "data": "synthetic",
            **({"experiment": "cache_latents: the un-warped views' latents are reused from the first epoch over the "
                              f"{len(data)}-image synthetic set (half of each step's VAE work skipped); NOT the bench of record"}
               if a.cache_latents else {}),
            "config": {"workload": f"{cfg_name}: T={a.tokens} R={a.res} K={a.top_k}/{a.candidates} {a.model} {image_size}^2 "
                                   f"{per_rank} img/rank x 2 views fp32",
                       "workload_detail": f"{a.model} architecture UNet+VAE (seeded synthetic weights), {image_size}x{image_size}, batch "
                                          f"{per_rank} images/rank/step x 2 views, T={a.tokens} tokens x {width}, R={a.res}, "
                                          f"top_k={a.top_k} of {a.candidates}, fp32 end to end",
                       "global_batch": global_batch, "images_per_rank": per_rank, "tokens": a.tokens, "embedding_dim": width,
                       "feature_upsample_res": a.res, "parallelism": f"dp{world}",
                       "kept_across_steps": "functions of the frozen weights and the fixed timestep only: transformed convolution "
                                            "filters, summed biases, the time embedding and the per-resnet offsets derived from it "
                                            "(ldm/fused.py); nothing that depends on an image, the noise or the embedding "
                                            "(latents are re-encoded every step: --cache-latents is a separate experiment line)"},
            # dominant kernel of the step by time: the Winograd conv of the frozen blocks, priced on the
            # fp32 matrix-core peak with the FLOPs it actually executes (direct-form FLOPs / 4)
            "roofline": {"kernel": f"skp_wino4_conv_c128_kernel<STATS={'true' if cv_folded else 'false'}, GNF={'true' if cv_folded else 'false'}> "
                                   f"(Winograd F(4x4,3x3) 3x3 conv, 128->128 ch at {min(image_size, 512)}^2, {cv_rows} rows: heaviest launch "
                                   "shape of the step IN THE FORM THE STEP LAUNCHES IT -- GroupNorm+SiLU folded into the patch load, block "
                                   "statistics in the epilogue; step_forms has the plain form of rounds 1-3 and the other shapes, "
                                   "conv_all_launches the time-weighted fraction over every Winograd launch of a step)",
                         "bound": "mfma", "achieved": cv_direct / 4 / cv_t / 1e12, "peak": F32_MATRIX_PEAK_TF,
                         "unit": "TFLOP/s", "frac": cv_direct / 4 / cv_t / 1e12 / F32_MATRIX_PEAK_TF,
                         "traffic": conv_traffic, "traffic_source": conv_src,
                         "launch_us": cv_t * 1e6, "algorithmic_flops": cv_direct / 4, "algorithmic_bytes": cv_bytes,
                         "direct_form_flops": cv_direct, "direct_form_equiv_tflops": cv_direct / cv_t / 1e12,
                         "rows_per_launch": cv_rows, "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                         "peak_note": "frac is against the NOMINAL 157.3 TF/s; mfma_issue_ceiling gives what back-to-back fp32 MFMAs "
                                      f"retire on this box ({'%.1f TF/s' % issue1 if issue1 else 'not measured'} at one wave per SIMD, which is what this kernel holds)",
                         "step_forms": cv_forms, "conv_all_launches": conv_all},
            # the north-star attention kernel (BASELINE metric: "fraction of the attention roofline")
            "roofline_attn_map": {"kernel": kt["fwd_kernel"] + " (fused up-res softmax map, forward)", "bwd_route": kt["bwd_route"],
                         "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": map_traffic, "traffic_source": map_src,
                         "launch_us": kt["fwd"] * 1e6, "algorithmic_bytes": fwd_bytes, "rows_per_launch": B,
                         "bwd_launch_us": kt["bwd"] * 1e6, "bwd_achieved": bwd_bytes / kt["bwd"] / 1e9,
                         "bwd_frac": bwd_bytes / kt["bwd"] / 1e9 / HBM_PEAK_GBS, "bwd_algorithmic_bytes": bwd_bytes,
                         "bwd_traffic": map_bwd_traffic,
                         "valu": (map_valu_roof(B, len(ldims or [0] * 4), (ldims or [(16, 1280, 8)])[0][2], a.tokens, a.res, kt["fwd"])
                                  if a.tokens <= 128 else None),
                         "bound_note": "the HBM fraction above is the contract's figure; the kernel is VALU-issue bound by design (the "
                                       "logits are up-sampled instead of the activations: 1.01x algorithmic traffic) -- `valu` prices it "
                                       "against the roof that binds it",
                         "reference_contraction_equiv_tflops": flops_equiv / kt["fwd"] / 1e12,
                         "f32_matrix_peak_tflops": F32_MATRIX_PEAK_TF},
            # flash self-attention of the 64^2 layers: bf16 matrix cores, every fp32 product = 6 bf16 products (three-term splits);
            # forward 2 tile products, backward (two kernels) 7 -- executed bf16 FLOPs against the dense bf16 peak, and the
            # fp32-equivalent rate (algorithmic FLOPs / time) next to the fp32 matrix peak it replaces
            "roofline_self_attn": {"kernel": "skp_fas_fwd_kernel<40> (+ K/V split pre-pass): flash self-attention forward, 64^2 layers "
                                             "(N=4096, 8 heads x 40), v_mfma_f32_16x16x32_bf16 on 3-term operand splits",
                                   "bound": "mfma", "achieved": 6 * sa_f / sa["fwd"] / 1e12, "peak": BF16_MATRIX_PEAK_TF,
                                   "unit": "TFLOP/s", "frac": 6 * sa_f / sa["fwd"] / 1e12 / BF16_MATRIX_PEAK_TF,
                                   "fp32_equiv_tflops": sa_f / sa["fwd"] / 1e12, "fp32_matrix_peak_tflops": F32_MATRIX_PEAK_TF,
                                   "traffic": None, "launch_us": sa["fwd"] * 1e6, "algorithmic_flops": sa_f,
                                   "bwd_us": sa["bwd"] * 1e6, "bwd_fp32_equiv_tflops": sa_b / sa["bwd"] / 1e12,
                                   "bwd_executed_tflops": 6 * (7.0 / 5.0) * sa_b / sa["bwd"] / 1e12,
                                   "bwd_frac": 6 * (7.0 / 5.0) * sa_b / sa["bwd"] / 1e12 / BF16_MATRIX_PEAK_TF,
                                   "rows_per_launch": B, "dtype": "f32 in/out/accumulate; products on bf16 MFMA, 3 x bf16 terms per operand"},
            # what the matrix pipe sustains on THIS box for back-to-back independent fp32 MFMAs (no loads, no VALU work):
            # the practical ceiling under the nominal peak the fractions above are quoted against
            "mfma_issue_ceiling": {"unit": "TFLOP/s", "nominal_peak": F32_MATRIX_PEAK_TF,
                                   "one_wave_per_simd": issue1, "two_waves_per_simd": issue2,
                                   "note": "skp_probe_mfma_f32; the Winograd conv kernels hold one wave per SIMD "
                                           "(288 accumulators), the attention kernels two"},
            "attention_roofline_frac": {"map_fwd_hbm": ach / HBM_PEAK_GBS, "map_bwd_hbm": bwd_bytes / kt["bwd"] / 1e9 / HBM_PEAK_GBS,
                                        "self_attn_fwd_fp32_equiv_over_fp32_peak": sa_f / sa["fwd"] / 1e12 / F32_MATRIX_PEAK_TF,
                                        "self_attn_bwd_fp32_equiv_over_fp32_peak": sa_b / sa["bwd"] / 1e12 / F32_MATRIX_PEAK_TF},
            "f32_instr": f32_instr,
            "traffic_live_kernels": sorted(live) if live else None,
            "cpu_baseline": cpu_stats, "verify": verify, "collective_check": coll,
            "loss": float(last[0]), "setup_s": t_build, "cpu_baseline_s": t_cpu,
            "launch_thread_cpu_ms_per_step": c_launch / a.steps * 1e3,
            "captured_step": ({"on": True, "what": "forward + losses + backward of a step replayed from one hipGraph (optimize.GraphedStep; inputs, "
                                                   "noise and affines refreshed in static buffers before every replay); all-reduce + Adam eager",
                               "group_sizes_captured": sorted(k for k, v in graphed.state.items() if v != "eager" and v.get("graph") is not None)}
                              if graphed is not None else {"on": False}),
            "step_from_idle": {"launch_thread_returns_ms": t_issue * 1e3, "gpu_done_ms": t_idle_total * 1e3}, "prewarm_steps": 1, "gemm_tunableop_file": bool(gemm_tuned),
            "weights_init": weights_init,
        }
        if line["roofline"]["traffic"]:
            line["roofline"]["hbm_gbs_at_traffic"] = line["roofline"]["traffic"] / cv_t / 1e9
        print(json.dumps(line), flush=True)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
