"""End-to-end GPU parity on a reduced-width SD-topology model: the batched/fused MI355X step
(`optimize.group_step`: both views in one batch, early exit, fused maps, on-device selection, fused
losses, autograd through the UNet) against the oracle's per-image reference-order CPU step."""
import copy

import pytest
import torch

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
    torch.testing.assert_close(c_gpu.grad.cpu(), gref, rtol=5e-3, atol=5e-5 * gref.abs().max().item())
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
