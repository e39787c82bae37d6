#!/usr/bin/env python
"""bench.py -- images/sec of the token-optimisation step (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one optimizer step of the reference loop (optimize.py:339-425) on this rank's share of
the global batch: `--images-per-rank` (default 4 = BASELINE config 2's batch_size on 1 GPU) synthetic
512x512 images, each = VAE-encode + hooked UNet forward of the image AND of its affine copy, fused map
reduction x2, on-device token selection, both losses, backward to the [1,77,768] embedding; then one
RCCL all-reduce(sum) of the gradient and an Adam step.  Weak scaling: per-rank work is fixed, the
global batch is images-per-rank * N.  value = N * images-per-rank * K / max-over-ranks seconds.

Rank 0 prints ONE JSON line with `roofline` (the fused attention-map kernel, measured live with
events on the launch stream) and, at N=1, `cpu_baseline` (the oracle's reference-order CPU step on
the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
F32_MATRIX_PEAK_TF = 157.3     # fp32 MFMA == fp32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images-per-rank", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--res", type=int, default=128, help="feature_upsample_res (R)")
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--model", default="sd15")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--cpu-image-size", type=int, default=512)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--kernel-iters", type=int, default=30)
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen find)")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--no-fuse-norms", action="store_true", help="A/B: keep ATen GroupNorm/SiLU in the frozen blocks")
    return ap.parse_args()


def map_kernel_roofline(ops, B, T, R, iters, device):
    """Time the fused map kernels alone at the bench shapes (SD-1.5 hooked layers: 3 x (16^2, C=1280) +
    1 x (32^2, C=640), 8 heads) with events on the CURRENT stream (the one the C-ABI launches on)."""
    g = torch.Generator(device="cpu").manual_seed(0)
    dims = [(16, 1280)] * 3 + [(32, 640)]
    qs = [torch.randn(B, s * s, C, generator=g).to(device) for s, C in dims]
    ks = [torch.randn(1, T, C, generator=g).to(device) for s, C in dims]
    H = 8
    scales = [(C // H) ** -0.5 for _, C in dims]
    S = [ops.qk_logits(q, k, H, sc) for q, k, sc in zip(qs, ks, scales)]
    sides = [s for s, _ in dims]
    M, lse = ops._map_fwd(S, sides, B, H, T, R)
    dM = torch.randn_like(M)
    dS = [torch.empty_like(s_) for s_ in S]
    from stablekeypoints_amd import _native as N
    sp, k1 = N.ptr_array([t.data_ptr() for t in S])
    dp, k2 = N.ptr_array([t.data_ptr() for t in dS])
    si, k3 = N.int_array(sides)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(4, N.lib().skp_attn_map_bwd_workspace(si, 4, B, H, min(T, 128), R)) // 4, device=device)

    def run_fwd():
        if T > 128:                                              # token-group path (several launches)
            return ops._map_fwd(S, sides, B, H, T, R)
        N.check(N.lib().skp_attn_map_fwd_f32(sp, si, 4, B, H, T, R, M.data_ptr(), lse.data_ptr(), st), "fwd")

    def run_bwd():
        if T > 128:
            return ops._map_bwd(S, dS, sides, B, H, T, R, dM, lse)
        N.check(N.lib().skp_attn_map_bwd_f32(sp, dp, si, 4, B, H, T, R, dM.data_ptr(), lse.data_ptr(), ws.data_ptr(), st), "bwd")

    out = {}
    for name, fn in (("fwd", run_fwd), ("bwd", run_bwd)):
        for _ in range(3):
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
    for _ in range(2):
        o = fwd()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        o = fwd()
    e1.record(); torch.cuda.synchronize()
    out["fwd"] = e0.elapsed_time(e1) / iters * 1e-3
    o = fwd()
    for _ in range(2):
        torch.autograd.grad(o, (q, k, v), w, retain_graph=True)
    e0.record()
    for _ in range(iters):
        torch.autograd.grad(o, (q, k, v), w, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    out["bwd"] = e0.elapsed_time(e1) / iters * 1e-3
    flops_fwd = 4.0 * Nq * Nq * d * B * H
    return out, flops_fwd, 2.5 * flops_fwd


def conv_roofline(ops, B, image_size, iters, device):
    """The Winograd F(4x4,3x3) conv kernel at the heaviest launch shape of the step: the VAE encoder's first-level
    ResnetBlock2D convs (128 -> 128 channels at image resolution, B rows).  Executed MFMA FLOPs = direct-form FLOPs / 4
    (36 multiplies per 4x4 output tile and channel pair instead of 144); algorithmic bytes = input + output + filter."""
    g = torch.Generator(device="cpu").manual_seed(2)
    ci = co = 128
    B = min(B, max(1, (2 ** 31 - 1) // (ci * image_size * image_size * 4)))   # one launch (the op chunks larger batches)
    x = torch.randn(B, ci, image_size, image_size, generator=g).to(device)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(device)
    U = ops._wino4_filters(w, False)
    fn = lambda: ops._conv3x3_f4_raw(x, U, None, co)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    direct = 2.0 * 9 * ci * co * B * image_size * image_size
    nbytes = 4 * (B * ci * image_size ** 2 + B * co * image_size ** 2 + 36 * ci * co)
    grid_threads = 8 * ((((B * (image_size // 4) ** 2 + 15) // 16) + 7) // 8) * (co // 128) * 256   # 128-channel form
    return t, direct, nbytes, grid_threads, B


def cpu_baseline(ldm_cpu, args):
    """Oracle reference-order CPU step (oracle/cpu_path.py) on a bounded sample: ONE image (2 UNet+VAE
    forwards with materialised attention + backward + Adam) at --cpu-image-size, after one untimed warm-up
    of the allocator at 1/4 size."""
    from oracle import cpu_path
    # 256 hardware threads thrash torch's CPU kernels (measured: 541 s/image at 256 threads); use 32.
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    size = args.cpu_image_size
    imgs = torch.rand(1, 3, size, size, generator=g)
    ctx = torch.randn(1, args.tokens, 768, generator=g)
    _, sec, n = cpu_path.optimize_embedding_cpu(ldm_cpu, imgs, ctx, steps=1, batch_size=1, R_up=args.res)
    return {"value": n / sec, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"1 optimizer step, batch 1, one {size}x{size} image (2 VAE+UNet forwards, backward, Adam), "
                      f"T={args.tokens}, R={args.res}, torch {torch.__version__} CPU fp32, {sec:.1f} s"}


def measured_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/*.json, separate
    FETCH_SIZE / WRITE_SIZE runs of tools/kbench.py at the same launch shape; FETCH_SIZE doubled per
    MI355X_MICROARCH.md 'HBM').  None when no profile is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), reverse=True)       # newest round / version first by name
    for path in files:
        try:
            k = json.load(open(path))["kernels"]
        except Exception:
            continue
        for name, c in k.items():
            if kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    return None


def main():
    a = parse()
    from stablekeypoints_amd import dist as D
    world, rank, local = D.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    if os.environ.get("SKP_BENCH_SINGLE_DEVICE") == "1":       # test hook: several ranks share cuda:0 (with gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # host threads: N ranks share the box's cores (weight init + the Python driver are the only CPU work)
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 8) // max(1, world))))
    from stablekeypoints_amd import ops, _native
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import SyntheticImages, default_args, group_step
    from stablekeypoints_amd.optimize_token import load_ldm
    _native.lib()

    t_build = time.time()
    ldm, controllers, _ = load_ldm("cpu", a.model, feature_upsample_res=a.res)      # seeded synthetic weights
    cpu_stats = None
    want_cpu = a.cpu_baseline == "on" or (a.cpu_baseline == "auto" and world == 1)
    if want_cpu and rank == 0:
        cpu_stats = cpu_baseline(ldm, a)
    # move the same instance to the GPU and install the fused hook (overrides the oracle's patch)
    ldm.to(dev)
    if a.miopen_find:
        torch.backends.cudnn.benchmark = True
    if a.channels_last:
        ldm.unet.to(memory_format=torch.channels_last)
        ldm.vae.to(memory_format=torch.channels_last)
    from stablekeypoints_amd import ptp_utils
    controller = ptp_utils.AttentionStore()
    controllers = {dev: controller}
    ptp_utils.register_attention_control(ldm.unet, controller, feature_upsample_res=a.res)
    ptp_utils.accelerate_cross_attention(ldm.unet)
    if not a.no_fuse_norms:
        from stablekeypoints_amd.ldm.fused import fuse_norms
        fuse_norms(ldm.unet)
        fuse_norms(ldm.vae)
    from stablekeypoints_amd import tuning
    gemm_tuned = tuning.enable()
    t_build = time.time() - t_build

    per_rank = a.images_per_rank
    global_batch = per_rank * world
    args = default_args(num_tokens=a.tokens, feature_upsample_res=a.res, batch_size=global_batch, device=str(dev),
                        image_size=a.image_size)
    data = SyntheticImages(n=max(16, per_rank * 2), size=a.image_size, seed=rank, device=dev)
    torch.manual_seed(1000 + rank)                                                    # per-rank augmentations/noise
    ctx = torch.randn(1, a.tokens, 768, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(True)
    opt = torch.optim.Adam([ctx], lr=args.lr)
    reducer = D.EmbeddingReducer(ctx, opt)
    transform = RandomAffineWithInverse(args.augment_degrees, args.augment_scale, args.augment_translate)

    cursor = 0

    def one_step():
        nonlocal cursor
        idx = [(cursor + i) % len(data) for i in range(per_rank)]
        cursor += per_rank
        images = torch.stack([data[i]["img"] for i in idx])
        out = group_step(ldm, images, ctx, args, controller, transform, denom=global_batch)
        reducer.step()
        return out

    one_step()                       # untimed pre-warm: MIOpen/hipBLASLt first-call solver selection, allocator growth
    for _ in range(a.warmup):
        one_step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = one_step()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        B = 2 * per_rank                                          # rows per fused-map launch (both views)
        kt, fwd_bytes, bwd_bytes, flops_equiv = map_kernel_roofline(ops, B, a.tokens, a.res, a.kernel_iters, dev)
        sa, sa_f, sa_b = self_attn_roofline(ops, B, max(3, a.kernel_iters // 6), dev)
        cv_t, cv_direct, cv_bytes, cv_grid, cv_rows = conv_roofline(ops, B, a.image_size, max(3, a.kernel_iters // 6), dev)
        ach = fwd_bytes / kt["fwd"] / 1e9
        value = global_batch * a.steps / elapsed
        line = {
            "metric": "images/sec for token-optimization step (SD1.5, 512^2, K=10 kpts)",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: SD1.5-architecture UNet+VAE (seeded synthetic weights), "
                                   f"{a.image_size}x{a.image_size}, batch {per_rank} images/rank/step x 2 views, "
                                   f"T={a.tokens} tokens, R={a.res}, top_k=10 of 25, fp32 end to end",
                       "global_batch": global_batch, "images_per_rank": per_rank, "tokens": a.tokens,
                       "feature_upsample_res": a.res, "parallelism": f"dp{world}"},
            # dominant kernel of the step by time (37 % of it): the Winograd conv of the frozen blocks, priced on the
            # fp32 matrix-core peak with the FLOPs it actually executes (direct-form FLOPs / 4)
            "roofline": {"kernel": f"skp_wino4_conv_c128_kernel (Winograd F(4x4,3x3) 3x3 conv, 128->128 ch at {a.image_size}^2, "
                                   f"{cv_rows} rows: heaviest launch shape of the step)",
                         "bound": "mfma", "achieved": cv_direct / 4 / cv_t / 1e12, "peak": F32_MATRIX_PEAK_TF,
                         "unit": "TFLOP/s", "frac": cv_direct / 4 / cv_t / 1e12 / F32_MATRIX_PEAK_TF,
                         "traffic": measured_traffic(f"skp_wino4_conv_c128_kernel@grid{cv_grid}"),
                         "launch_us": cv_t * 1e6, "algorithmic_flops": cv_direct / 4, "algorithmic_bytes": cv_bytes,
                         "direct_form_flops": cv_direct, "direct_form_equiv_tflops": cv_direct / cv_t / 1e12,
                         "rows_per_launch": cv_rows, "dtype": "f32 (v_mfma_f32_16x16x4_f32)"},
            "roofline_attn_map": {"kernel": "skp_attn_map_fwd_kernel<80,0> (fused up-res softmax map, forward)",
                         "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": measured_traffic("skp_attn_map_fwd_kernel"),
                         "launch_us": kt["fwd"] * 1e6, "algorithmic_bytes": fwd_bytes, "rows_per_launch": B,
                         "bwd_launch_us": kt["bwd"] * 1e6, "bwd_achieved": bwd_bytes / kt["bwd"] / 1e9,
                         "reference_contraction_equiv_tflops": flops_equiv / kt["fwd"] / 1e12,
                         "f32_matrix_peak_tflops": F32_MATRIX_PEAK_TF},
            "roofline_self_attn": {"kernel": "skp_self_attn_fwd_kernel<5,2> (flash self-attention, 64^2 layers)",
                                   "bound": "mfma", "achieved": sa_f / sa["fwd"] / 1e12, "peak": F32_MATRIX_PEAK_TF,
                                   "unit": "TFLOP/s", "frac": sa_f / sa["fwd"] / 1e12 / F32_MATRIX_PEAK_TF,
                                   "traffic": None, "launch_us": sa["fwd"] * 1e6, "algorithmic_flops": sa_f,
                                   "bwd_us": sa["bwd"] * 1e6, "bwd_achieved": sa_b / sa["bwd"] / 1e12,
                                   "rows_per_launch": B, "dtype": "f32 (v_mfma_f32_32x32x2_f32)"},
            "cpu_baseline": cpu_stats,
            "loss": float(last[0]), "build_s": t_build, "prewarm_steps": 1, "gemm_tunableop_file": bool(gemm_tuned),
        }
        if line["roofline"]["traffic"]:
            line["roofline"]["hbm_gbs_at_traffic"] = line["roofline"]["traffic"] / cv_t / 1e9
        print(json.dumps(line), flush=True)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
