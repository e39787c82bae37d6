"""Round 6: whole-step coverage of two reference options (`--num_subjects > 1`; `load_ldm`'s default
`feature_upsample_res=256`) and the kernels added this round."""
import pytest
import torch
from _tol import assert_grad_close

from oracle import ref_path as R

pytestmark = pytest.mark.gpu


def _oracle_step(cpu, images, ctx, thetas, noise, Rup, n_cand, top_k, sigma, num_subjects=1):
    """The oracle's reference-order CPU step, image by image (optimize.py:347-425): summed gradient / n, mean losses."""
    from oracle import cpu_path
    n = images.shape[0]
    store = R.OracleStore()
    cpu_path.register_reference_hook(cpu.unet, store, Rup)
    gref, sels, sharp_m, equiv_m = torch.zeros_like(ctx), [], 0.0, 0.0
    for i in range(n):
        c_ref = ctx.clone().requires_grad_(True)
        loss, sharp, equiv, sel, _, _ = cpu_path.image_step(cpu, images[i:i + 1], c_ref, store, thetas[i:i + 1], noise[i:i + 1],
                                                            noise[n + i:n + i + 1], furthest_point_num_samples=n_cand, top_k=top_k,
                                                            sigma=sigma, num_subjects=num_subjects)
        loss.backward()
        gref += c_ref.grad / n
        sels.append(sel)
        sharp_m += sharp.item() / n
        equiv_m += equiv.item() / n
    return gref, sels, sharp_m, equiv_m


def test_group_step_two_subjects_vs_oracle():
    """`--num_subjects 2` (main.py; eval.py:62-81 `find_k_max_pixels` + `mask_radius`, optimize.py:166-179 sharpening towards the
    two maxima) through the FUSED training node: `group_step` -> `ops.MapLossesFn` with the batched statistics of all images
    reshaped per subject (`reshape(ns, n, T)`), against the oracle's per-image reference-order step with num_subjects = 2."""
    from test_e2e_gpu import _setup
    from stablekeypoints_amd import ops
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import group_step
    # 512^2 images on the reduced-width tree: hooked layers 16^2 x 3 + 32^2 at R = 128, the shapes whose step takes the fused node
    # (ops.map_bwd_sparse_supported: column-sweep backward) -- at the tiny default (R = 32) the step runs image by image
    ldm, controllers, cpu, images, ctx, noise, args = _setup(R_up=128, T=16, n=2, size=512)
    args.num_subjects = 2
    n = images.shape[0]
    dev, controller = next(iter(controllers.items()))
    thetas = torch.cat([R.affine_matrix(11.0, 0.87, (0.13, -0.21)), R.affine_matrix(-9.0, 0.93, (-0.2, 0.1))])
    gref, sels, sharp_ref, equiv_ref = _oracle_step(cpu, images, ctx, thetas, noise, 128, 8, 4, args.sigma, num_subjects=2)
    _, _, sharp_1, _ = _oracle_step(cpu, images, ctx, thetas, noise, 128, 8, 4, args.sigma, num_subjects=1)
    assert abs(sharp_ref - sharp_1) > 1e-3 * abs(sharp_1), "the two-subject target must differ from the one-subject one on this seed"
    seen = []
    real_stats = ops.token_stats

    def spy_stats(M, num_subjects=1, **kw):                       # the fused node's batched statistics: all images in one launch
        seen.append((int(M.shape[0]), int(num_subjects)))
        return real_stats(M, num_subjects=num_subjects, **kw)
    ops.token_stats = spy_stats
    try:
        c_gpu = ctx.clone().cuda().requires_grad_(True)
        tr = RandomAffineWithInverse()
        loss, eq, sh = group_step(ldm, images, c_gpu, args, controller, tr, denom=n, noise=noise.cuda(), thetas=thetas)
    finally:
        ops.token_stats = real_stats
    T = ctx.shape[1]
    assert (n * T, 2) in seen, seen                                 # ns = 2 went through the batched route of ops.MapLossesFn
    # the selection with two subjects, image by image on the same maps (optimize.image_losses: the per-image form of the node)
    from stablekeypoints_amd import ptp_utils
    from stablekeypoints_amd._maps import collect_maps_batched
    from stablekeypoints_amd.optimize import image_losses
    with torch.no_grad():
        both = torch.cat([images.cuda(), tr(images.cuda(), theta=thetas)])
        ptp_utils.find_pred_noise(ldm, both, ctx.cuda(), device=dev, noise=noise.cuda(), early_exit=True, controllers=controllers)
        maps = collect_maps_batched(controller)
    for i in range(n):
        _, _, sel_g = image_losses(maps[i], maps[n + i], thetas[i].reshape(-1).tolist(), args)
        assert torch.equal(sel_g.cpu(), sels[i]), f"image {i}: selected tokens differ from the reference's"
    assert abs(sh.item() - sharp_ref) < 1e-3 * abs(sharp_ref)
    assert abs(eq.item() - equiv_ref) < 2e-3 * abs(equiv_ref)
    assert_grad_close(c_gpu.grad, gref, "two subjects, tiny tree")


def test_sd15_step_at_load_ldm_default_res_256_vs_oracle(sd15_cpu):
    """One full-width SD-1.5 step at `feature_upsample_res=256` -- `load_ldm`'s own default (optimize_token.py:24) -- 512^2,
    one image x 2 views, T = 77, K = 10 of 25: hooked layers 16^2 x 3 + 32^2 up-sampled 16x / 8x (the column-sweep map backward
    serves 4x / 8x only, so this step runs the general dense-gradient route and R = 256 tiles of the forward) against the
    oracle's reference-order CPU step."""
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import default_args, group_step
    from stablekeypoints_amd.optimize_token import load_ldm
    torch.set_num_threads(min(32, torch.get_num_threads()))
    Rup, T, n_cand, top_k = 256, 77, 25, 10
    ldm, controllers, _ = load_ldm("cuda", "sd15")                       # feature_upsample_res left at its default
    dev, controller = next(iter(controllers.items()))
    cpu = sd15_cpu
    g = torch.Generator().manual_seed(11)
    images = torch.rand(1, 3, 512, 512, generator=g)
    ctx = torch.randn(1, T, 768, generator=g) * 5.0
    noise = torch.randn(2, 4, 64, 64, generator=g)
    thetas = R.affine_matrix(-8.0, 0.9, (0.12, -0.1))
    args = default_args(num_tokens=T, feature_upsample_res=Rup, furthest_point_num_samples=n_cand, top_k=top_k, batch_size=1)
    gref, sels, sharp_ref, equiv_ref = _oracle_step(cpu, images, ctx, thetas, noise, Rup, n_cand, top_k, args.sigma)
    c_gpu = ctx.clone().cuda().requires_grad_(True)
    from stablekeypoints_amd import ops
    seen = []
    real = ops._map_fwd

    def spy(S, sides, B, H, T_, R_, *a, **k):
        seen.append((tuple(sides), R_))
        return real(S, sides, B, H, T_, R_, *a, **k)
    ops._map_fwd = spy
    try:
        loss, eq, sh = group_step(ldm, images, c_gpu, args, controller, RandomAffineWithInverse(), denom=1, noise=noise.cuda(), thetas=thetas)
    finally:
        ops._map_fwd = real
    assert seen == [((16, 16, 16, 32), 256)], seen
    assert abs(sh.item() - sharp_ref) < 1e-3 * abs(sharp_ref)
    assert abs(eq.item() - equiv_ref) < 2e-3 * abs(equiv_ref)
    assert_grad_close(c_gpu.grad, gref, "sd15 512^2, R = 256 (load_ldm default)")


def test_graphed_step_replays_the_eager_step():
    """`optimize.GraphedStep`: forward + losses + backward of a group captured in a hipGraph and replayed with new images, noise and
    affines per call (static buffers; the inverse affines read from device memory by skp_losses_fwd_dev_f32).  Reduced-width tree at
    512^2 / R = 128 (the shapes whose step takes the fused node), 1 image per group: five steps, each against the eager
    `group_step` on the same inputs -- losses and the accumulated embedding gradient (same kernels; the libraries may pick their
    algorithms per stream, hence 1e-5 of the maximum instead of bit equality) -- then with the noise and the thetas drawn by the
    step itself under one seed: the same draws in both modes."""
    from test_e2e_gpu import _setup
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import GraphedStep, group_step
    ldm, controllers, cpu, images, ctx, noise, args = _setup(R_up=128, T=16, n=2, size=512)
    del cpu
    dev, controller = next(iter(controllers.items()))
    g = torch.Generator().manual_seed(5)
    tr_a, tr_b = RandomAffineWithInverse(15, (0.8, 1.0), (0.25, 0.25)), RandomAffineWithInverse(15, (0.8, 1.0), (0.25, 0.25))
    c_e = ctx.clone().cuda().requires_grad_(True)
    c_g = ctx.clone().cuda().requires_grad_(True)
    graphed = GraphedStep(ldm, c_g, args, controller, tr_b, denom=1, warmup=2)
    for step in range(5):
        img = torch.rand(1, 3, 512, 512, generator=g)
        nz = torch.randn(2, 4, 64, 64, generator=g).cuda()
        th = R.affine_matrix(float(torch.rand(1, generator=g)) * 20 - 10, 0.85 + 0.1 * float(torch.rand(1, generator=g)), (0.1, -0.05 * step))
        le = group_step(ldm, img, c_e, args, controller, tr_a, denom=1, noise=nz, thetas=th)
        lg = [v.clone() for v in graphed(img, noise=nz, thetas=th)]
        for a_, b_ in zip(le, lg):
            assert abs(a_.item() - b_.item()) <= 1e-5 * abs(a_.item()) + 1e-9, (step, a_.item(), b_.item())
        assert_grad_close(c_g.grad, c_e.grad, f"graphed vs eager, step {step}", tol=1e-5)     # gradients ACCUMULATE over the five steps
    st = graphed.state[1]
    assert st != "eager" and st["graph"] is not None and st["calls"] == 2, "steps 3-5 must have been replays"
    # draws made by the step itself: same generators, same order
    c_e.grad = None
    c_g.grad.zero_()
    img = torch.rand(1, 3, 512, 512, generator=g)
    torch.manual_seed(77)
    le = group_step(ldm, img, c_e, args, controller, tr_a, denom=1)
    torch.manual_seed(77)
    lg = [v.clone() for v in graphed(img)]
    assert torch.equal(tr_a.last_theta_host, tr_b.last_theta_host)
    assert abs(le[0].item() - lg[0].item()) <= 1e-5 * abs(le[0].item())
    assert_grad_close(c_g.grad, c_e.grad, "graphed vs eager, own draws", tol=1e-5)
    # a group size whose step does not take the fused node stays eager, with one warning
    ldm2, controllers2, _, images2, ctx2, noise2, args2 = _setup()            # R = 32: image-by-image losses
    dev2, controller2 = next(iter(controllers2.items()))
    c2 = ctx2.clone().cuda().requires_grad_(True)
    g2 = GraphedStep(ldm2, c2, args2, controller2, RandomAffineWithInverse(), denom=2, warmup=1)
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(3):
            g2(images2)
    assert g2.state[2] == "eager" and sum("stay on eager" in str(x.message) for x in w) == 1


def test_vae_tail_composed_convolution_vs_module():
    """`AutoencoderKL.encode` on the fused path: `quant_conv(conv_out(silu(GroupNorm(h))))` as ONE Winograd convolution with the
    composed, zero-padded filter (ldm/fused.py: _vae_encode) against the module's own forward in fp64 on the same weights: posterior
    mean and log-variance; the composition follows the weights' versions."""
    import copy
    from stablekeypoints_amd.ldm.fused import fuse_norms
    from stablekeypoints_amd.ldm.vae import AutoencoderKL
    torch.manual_seed(3)
    ref = AutoencoderKL(block_out_channels=(32, 64, 64, 64)).eval()
    for p_ in ref.parameters():
        p_.requires_grad_(False)
    with torch.no_grad():
        ref.quant_conv.weight.normal_(0, 0.5); ref.quant_conv.bias.normal_(0, 0.5); ref.encoder.conv_out.bias.normal_(0, 0.5)
    gpu = copy.deepcopy(ref).cuda()
    assert fuse_norms(gpu) > 0 and "encode" in gpu.__dict__
    x = torch.rand(2, 3, 128, 128) * 2 - 1
    with torch.no_grad():
        want = ref.double().encode(x.double())["latent_dist"]
        got = gpu.encode(x.cuda())["latent_dist"]
        assert "_skp_tail" in gpu.__dict__, "the composed tail was not taken"
        for a, b in ((got.mean, want.mean), (got.logvar, want.logvar)):
            torch.testing.assert_close(a.double().cpu(), b, rtol=1e-4, atol=2e-5 * b.abs().max().item())
        gpu.quant_conv.bias.add_(1.0)                                        # version bump: the composition must follow
        got2 = gpu.encode(x.cuda())["latent_dist"]
        torch.testing.assert_close(got2.mean, got.mean + 1.0, rtol=1e-5, atol=1e-5)


def test_conv_in_block_statistics_feed_the_first_group_norm():
    """The VAE's conv_in (3 -> 128 at image resolution) with block statistics in its epilogue (skp_conv3x3_small_stats_f32: {mean,
    sum of squared deviations} per 512 consecutive pixels, sums about the bias, DPP wave reduction): the output is bit-identical
    to the plain launch, the block moments match torch, and the GroupNorm that consumes them (folded into the next convolution's
    patch load) gives what the statistics pass over the activation gives."""
    from stablekeypoints_amd import ops
    g = torch.Generator().manual_seed(12)
    B, co, H, W = 4, 128, 512, 512
    x = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).cuda()
    w = (torch.randn(co, 3, 3, 3, generator=g) * 0.3).cuda()
    b = (torch.randn(co, generator=g) * 3.0).cuda()                  # means far from zero: the pivot matters
    assert ops.N.lib().skp_conv3x3_small_stats_blocks(B, 3, co, H, W) == H * W // 512
    assert ops.N.lib().skp_conv3x3_small_stats_blocks(1, 3, co, 64, 64) == 0 and ops.N.lib().skp_conv3x3_small_stats_blocks(B, 4, co, H, W) == 0
    y0 = ops.conv3x3_small(x, w, b)
    y = ops.conv3x3_small(x, w, b, want_stats=True)
    assert torch.equal(y, y0) and getattr(y, "_skp_blocks", None) is not None
    stats, nblk, pix = y._skp_blocks
    assert (nblk, pix) == (H * W // 512, 512) and stats.shape == (B, co, nblk, 2)
    blk = y.double().reshape(B, co, nblk, 512)
    torch.testing.assert_close(stats[..., 0].double(), blk.mean(-1), rtol=1e-5, atol=1e-5)
    m2 = ((blk - blk.mean(-1, keepdim=True)) ** 2).sum(-1)
    torch.testing.assert_close(stats[..., 1].double(), m2, rtol=2e-4, atol=1e-3)
    norm = torch.nn.GroupNorm(32, co, eps=1e-6).cuda()
    with torch.no_grad():
        norm.weight.normal_(1.0, 0.2, generator=None); norm.bias.normal_(0.0, 0.2)
        a = ops.group_norm_silu(y, norm)                               # statistics from the blocks
        b_ = ops.group_norm_silu(y0, norm)                             # statistics by their own pass
        ref = torch.nn.functional.silu(torch.nn.functional.group_norm(y0.double(), 32, norm.weight.double(), norm.bias.double(), 1e-6))
    torch.testing.assert_close(a.double(), ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(a, b_, rtol=1e-4, atol=1e-5)
