"""End-to-end GPU parity on a reduced-width SD-topology model: the batched/fused MI355X step
(`optimize.group_step`: both views in one batch, early exit, fused maps, on-device selection, fused
losses, autograd through the UNet) against the oracle's per-image reference-order CPU step."""
import copy

import pytest
import torch
from _tol import assert_grad_close

pytestmark = pytest.mark.gpu


def _setup(R_up=32, T=16, n=2, size=128):
    from stablekeypoints_amd.optimize_token import load_ldm
    from stablekeypoints_amd.optimize import default_args
    ldm, controllers, _ = load_ldm("cuda", "tiny", feature_upsample_res=R_up)
    cpu, _, _ = load_ldm("cpu", "tiny", feature_upsample_res=R_up)          # same seed => same weights
    g = torch.Generator().manual_seed(0)
    images = torch.rand(n, 3, size, size, generator=g)
    ctx = torch.randn(1, T, 768, generator=g)
    noise = torch.randn(2 * n, 4, size // 8, size // 8, generator=g)
    args = default_args(num_tokens=T, feature_upsample_res=R_up, furthest_point_num_samples=8, top_k=4,
                        batch_size=n, device="cuda")
    return ldm, controllers, cpu, images, ctx, noise, args


def test_group_step_matches_oracle_reference_order():
    from oracle import cpu_path, ref_path as R
    from stablekeypoints_amd import ptp_utils
    from stablekeypoints_amd._maps import collect_maps_batched
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import group_step
    ldm, controllers, cpu, images, ctx, noise, args = _setup()
    n = images.shape[0]
    dev, controller = next(iter(controllers.items()))
    thetas = torch.cat([R.affine_matrix(11.0, 0.87, (0.13, -0.21)), R.affine_matrix(-9.0, 0.93, (-0.2, 0.1))])
    # ---- oracle: per image, full forwards, materialised stores, python selection -------------
    store = R.OracleStore()
    assert cpu_path.register_reference_hook(cpu.unet, store, 32) == 18
    c_ref = ctx.clone().requires_grad_(True)
    ref = []
    for i in range(n):
        loss, sharp, equiv, sel, am, am_t = cpu_path.image_step(
            cpu, images[i:i + 1], c_ref, store, thetas[i:i + 1], noise[i:i + 1], noise[n + i:n + i + 1],
            furthest_point_num_samples=8, top_k=4, sigma=args.sigma)
        (loss / n).backward()
        ref.append((sharp.item(), equiv.item(), sel, am.detach(), am_t.detach()))
    # ---- maps of the fused path (no grad) ------------------------------------------------------
    c_gpu = ctx.clone().cuda().requires_grad_(True)
    tr = RandomAffineWithInverse()
    with torch.no_grad():
        both = torch.cat([images.cuda(), tr(images.cuda(), theta=thetas)])
        ptp_utils.find_pred_noise(ldm, both, c_gpu, device=dev, noise=noise.cuda(), early_exit=True, controllers=controllers)
        maps = collect_maps_batched(controller)
    for i in range(n):
        torch.testing.assert_close(maps[i].cpu(), ref[i][3], rtol=1e-3, atol=1e-6)
        torch.testing.assert_close(maps[n + i].cpu(), ref[i][4], rtol=1e-3, atol=1e-6)
    # ---- the step itself ------------------------------------------------------------------------
    loss, eq, sh = group_step(ldm, images, c_gpu, args, controller, tr, denom=n, noise=noise.cuda(), thetas=thetas)
    sh_ref = sum(r[0] for r in ref) / n
    eq_ref = sum(r[1] for r in ref) / n
    assert abs(sh.item() - sh_ref) < 1e-3 * abs(sh_ref)
    assert abs(eq.item() - eq_ref) < 2e-3 * abs(eq_ref)
    gref = c_ref.grad
    assert_grad_close(c_gpu.grad, gref, "test_e2e_gpu.py#1")
    assert len(controller.step_store["attn"]) == 0


def test_full_forward_equals_early_exit_and_public_api():
    """`run_and_find_attn` / `collect_maps` public API: early exit is result-identical; `indices` and the
    bilinear `upsample_res` path; materialised compat store gives the same map."""
    from stablekeypoints_amd import ptp_utils
    ldm, controllers, cpu, images, ctx, noise, args = _setup()
    dev = next(iter(controllers))
    c = ctx.cuda()
    img = images[:1].cuda()
    with torch.no_grad():
        kw = dict(device=dev, controllers=controllers, noise=noise[:1].cuda(), layers=[0, 1, 2, 3])
        a = ptp_utils.run_and_find_attn(ldm, img, c, upsample_res=-1, early_exit=True, **kw)[0]
        b = ptp_utils.run_and_find_attn(ldm, img, c, upsample_res=-1, early_exit=False, **kw)[0]
        assert a.shape == (ctx.shape[1], 32, 32)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)     # early exit: same values (library kernels may reorder sums)
        idx = torch.tensor([3, 0, 7])
        up = ptp_utils.run_and_find_attn(ldm, img, c, upsample_res=64, indices=idx, **kw)[0]
        ref = torch.nn.functional.interpolate(a[idx.cuda()][None], size=(64, 64), mode="bilinear", align_corners=False)[0]
        torch.testing.assert_close(up, ref, rtol=1e-5, atol=1e-7)
        controllers[dev].materialize = True
        m = ptp_utils.run_and_find_attn(ldm, img, c, upsample_res=-1, **kw)[0]
        controllers[dev].materialize = False
        torch.testing.assert_close(m, a, rtol=1e-4, atol=1e-7)
        top = ptp_utils.find_top_k_gaussian(a, 8, sigma=2.0)
        fps = ptp_utils.furthest_point_sampling(a, 4, top)
    from oracle import ref_path as R
    assert torch.equal(top.cpu(), R.find_top_k_gaussian(a.cpu(), 8, sigma=2.0))
    assert torch.equal(fps.cpu(), R.furthest_point_sampling(a.cpu(), 4, top.cpu()))


def test_optimize_embedding_runs_and_decreases_loss():
    from stablekeypoints_amd.optimize import optimize_embedding, default_args
    from stablekeypoints_amd.optimize_token import load_ldm
    ldm, controllers, n = load_ldm("cuda", "tiny", feature_upsample_res=32)
    args = default_args(num_tokens=16, feature_upsample_res=32, furthest_point_num_samples=8, top_k=4, batch_size=2,
                        num_steps=3, image_size=128, max_len=4, device="cuda", log_interval=0)
    torch.manual_seed(0)
    ctx0 = torch.randn(1, 16, 768)
    out = optimize_embedding(ldm, args, controllers, n, context=ctx0.clone())
    assert out.shape == (1, 16, 768) and not out.requires_grad and torch.isfinite(out).all()
    step = (out.cpu() - ctx0).abs().max().item()
    assert 0 < step <= 3 * 5e-3 * 1.01                                     # 3 Adam steps of lr 5e-3


def test_g8_fused_step_vs_reference_driver_golden(golden=None):
    """The fused MI355X step against golden values produced by the REFERENCE's own driver functions
    (run_and_find_attn, find_top_k_gaussian, furthest_point_sampling, sharpening/equivariance loss, backward)
    on the same reduced-width model (tests/golden/g8_reference_step_tiny.npz)."""
    import numpy as np, os
    from oracle.fixtures import TINY_CASE as tc, seeded
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import default_args, group_step
    from stablekeypoints_amd.optimize_token import load_ldm
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_reference_step_tiny.npz")))
    ldm, controllers, _ = load_ldm("cuda", "tiny", feature_upsample_res=tc["R"])
    dev, controller = next(iter(controllers.items()))
    image = torch.rand(1, 3, tc["size"], tc["size"], generator=torch.Generator().manual_seed(tc["seed"]))
    ctx = seeded((1, tc["T"], 768), tc["seed"] + 1).cuda().requires_grad_(True)
    args = default_args(num_tokens=tc["T"], feature_upsample_res=tc["R"], furthest_point_num_samples=tc["n_cand"],
                        top_k=tc["top_k"], sigma=tc["sigma"], batch_size=1)
    loss, eq, sh = group_step(ldm, image, ctx, args, controller, RandomAffineWithInverse(), denom=1,
                              noise=torch.from_numpy(g["noise"]).cuda(), thetas=torch.from_numpy(g["theta"]))
    assert abs(sh.item() - float(g["sharp"])) < 1e-3 * abs(float(g["sharp"]))
    assert abs(eq.item() - float(g["equiv"])) < 2e-3 * abs(float(g["equiv"]))
    ref = torch.from_numpy(g["context_grad"])
    assert_grad_close(ctx.grad, ref, "test_e2e_gpu.py#2")


def test_g9_augmented_inference_and_keypoints_vs_reference_golden():
    """run_image_with_context_augmented + arg-max keypoints against the reference's eval.py output (G9)."""
    import numpy as np, os
    from oracle.fixtures import TINY_CASE as tc, seeded
    from stablekeypoints_amd.eval import run_image_with_context_augmented
    from stablekeypoints_amd.keypoint_regressor import keypoints_from_maps
    from stablekeypoints_amd.optimize_token import load_ldm
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_reference_augmented_tiny.npz")))
    ldm, controllers, _ = load_ldm("cuda", "tiny", feature_upsample_res=tc["R"])
    image = torch.rand(1, 3, tc["size"], tc["size"], generator=torch.Generator().manual_seed(tc["seed"]))[0]
    ctx = seeded((1, tc["T"], 768), tc["seed"] + 1).cuda()
    maps = run_image_with_context_augmented(
        ldm, image.permute(1, 2, 0).numpy(), ctx, torch.from_numpy(g["indices"]), device="cuda", layers=[0, 1, 2, 3],
        augmentation_iterations=tc["aug_iters"], augment_degrees=30, augment_scale=(0.9, 1.1),
        augment_translate=(0.1, 0.1), controllers=controllers, num_gpus=1, upscale_size=tc["upscale"],
        thetas=torch.from_numpy(g["thetas"]), noise=torch.from_numpy(g["noise"]).cuda())
    ref = torch.from_numpy(g["maps"])
    assert maps.shape == ref.shape
    torch.testing.assert_close(maps.cpu(), ref, rtol=1e-3, atol=1e-6)
    kp = keypoints_from_maps(maps, "argmax").cpu()
    assert torch.equal(kp, torch.from_numpy(g["keypoints"]))               # final keypoint locations, bit-exact


def test_find_best_indices_runs():
    from stablekeypoints_amd.keypoint_regressor import find_best_indices
    from stablekeypoints_amd.optimize import default_args
    from stablekeypoints_amd.optimize_token import load_ldm
    ldm, controllers, n = load_ldm("cuda", "tiny", feature_upsample_res=32)
    args = default_args(num_tokens=16, feature_upsample_res=32, furthest_point_num_samples=8, top_k=4, image_size=128,
                        max_len=6, device="cuda", num_indices=6)
    idx = find_best_indices(ldm, torch.randn(1, 16, 768, generator=torch.Generator().manual_seed(2)).cuda(), args,
                            controllers, n)
    assert idx.shape == (4,) and idx.dtype == torch.int64 and len(set(idx.tolist())) == 4 and idx.max() < 16


def test_bench_two_ranks_one_gpu_gloo():
    """bench.py through torch.distributed.run with WORLD_SIZE=2 (both ranks on cuda:0, gloo instead of RCCL):
    barrier / max-over-ranks timing / gradient all-reduce / single JSON line from rank 0."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SKP_DIST_BACKEND="gloo", SKP_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(root, "bench.py"), "--gpus", "2", "--model", "tiny",
           "--image-size", "128", "--res", "32", "--tokens", "16", "--steps", "2", "--warmup", "1", "--cpu-baseline", "off",
           "--kernel-iters", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["scaling"] == "weak" and d["steps"] == 2
    assert d["collective_check"]["ok"] and d["collective_check"]["ranks_seen"] == 2
    assert d["collective_check"]["embedding_identical_on_all_ranks"] is True
    # strong scaling: the global batch stays at --global-batch, each rank takes half
    cmd = cmd[:cmd.index("--master-port") + 1] + [str(29900 + os.getpid() % 90)] + cmd[cmd.index("--master-port") + 2:]
    out = subprocess.run(cmd + ["--scaling", "strong", "--global-batch", "4"], capture_output=True, text=True, timeout=300,
                         env=env, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 4 and d["config"]["images_per_rank"] == 2


def test_sd_shapes_maps_with_winograd_convs_match_library_convs():
    """SD-1.5 architecture at 256^2 (BASELINE configs[0] shapes): the [T,R,R] maps and the context gradient computed
    with the Winograd conv3x3 kernels (F(4x4,3x3) / F(2x2,3x3)) against the same step with the library convolutions.
    Tolerance: maps rtol 1e-3 (north_star)."""
    from stablekeypoints_amd import ops, ptp_utils
    from stablekeypoints_amd.optimize_token import load_ldm
    ldm, controllers, n = load_ldm("cuda:0", "sd15", feature_upsample_res=64)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    images = torch.rand(2, 3, 256, 256, generator=g).to(dev)
    noise = torch.randn(2, 4, 32, 32, generator=g).to(dev)
    # context scaled so that the token softmax is clearly non-uniform (max/mean ~3.5) while the random-weight network is
    # still well conditioned: at this scale two runs of the LIBRARY path differ by ~1e-5 (its own nondeterminism), the
    # Winograd kernels by 1-2e-5; at 8x this scale the library path already disagrees with itself by 30 %.
    ctx0 = torch.randn(1, 77, 768, generator=g) * 5.0
    out = {}
    prev = ops.CONV3X3_MODE
    try:
        for mode in ("lib", "f4", "f2"):
            ops.CONV3X3_MODE = mode
            context = ctx0.clone().to(dev).requires_grad_(True)
            maps = ptp_utils.run_and_find_attn(ldm, images, context, layers=[0, 1, 2, 3], noise_level=-1, from_where=["up_cross"],
                                               upsample_res=64, device=dev, controllers=controllers, noise=noise)
            m = maps if torch.is_tensor(maps) else torch.stack(list(maps))
            (m * torch.linspace(0, 1, m.numel(), device=dev).reshape(m.shape)).sum().backward()
            out[mode] = (m.detach().clone(), context.grad.detach().clone())
    finally:
        ops.CONV3X3_MODE = prev
    ref_maps = out["lib"][0]
    print("map max/mean", (ref_maps.max() / ref_maps.mean()).item(), "grad max", out["lib"][1].abs().max().item())
    assert ref_maps.max() > 2.5 * ref_maps.mean()               # the comparison has teeth: maps are not uniform
    for mode in ("f4", "f2"):
        print(mode, "max rel map diff", ((out[mode][0] - ref_maps).abs() / ref_maps.abs().clamp_min(1e-6)).max().item())
        torch.testing.assert_close(out[mode][0], out["lib"][0], rtol=1e-3, atol=1e-6)
        gref = out["lib"][1]
        torch.testing.assert_close(out[mode][1], gref, rtol=5e-3, atol=1e-3 * gref.abs().max().item())
