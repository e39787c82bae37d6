"""Pin the CPU oracle (oracle/ref_path.py) to golden vectors captured from the REFERENCE's own
functions (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ref_path as R
from oracle.fixtures import (HOOK_CASES, STACK_CASE, SEL_CASE, E2E_CASE, attention_weights, seeded,
                             selection_maps, sharp_entropy_maps)

TOL = dict(rtol=2e-5, atol=2e-7)


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name", list(HOOK_CASES))
def test_g1_hook(golden, name):
    g = golden("g1_hook.npz")
    c = HOOK_CASES[name]
    w = attention_weights(c["C"], c["ctx_dim"], c["seed"])
    x = seeded((c["B"], c["s"] ** 2, c["C"]), c["seed"] + 100)
    ctx = seeded((c["B"], c["T"], c["ctx_dim"]), c["seed"] + 200)
    store = R.OracleStore()
    out = R.hooked_attention(x, ctx, *w, c["heads"], store, c["R"])
    assert len(store.step_store["attn"]) == 1
    p_up = store.step_store["attn"][0]
    assert p_up.shape == (c["B"] * c["heads"], c["R"] ** 2, c["T"])
    if c["full"]:
        torch.testing.assert_close(p_up, t(g[name + "/p_up"]), **TOL)
        torch.testing.assert_close(out, t(g[name + "/out"]), rtol=1e-4, atol=1e-5)
    else:
        torch.testing.assert_close(p_up.reshape(-1)[:: c["stride"]], t(g[name + "/p_up_strided"]), rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(out.reshape(-1)[:: c["stride"]], t(g[name + "/out_strided"]), rtol=1e-4, atol=1e-5)
        assert abs(p_up.double().sum().item() - float(g[name + "/p_up_sum"])) < 1e-2
    # self-attention (context=None) never stores  (ptp_utils.py:509)
    ws = attention_weights(c["C"], c["C"], c["seed"] + 1000)
    out_self = R.hooked_attention(x, None, *ws, c["heads"], store, c["R"])
    assert len(store.step_store["attn"]) == 1
    torch.testing.assert_close(out_self.reshape(-1)[::7], t(g[name + "/out_self_strided"]), rtol=1e-4, atol=1e-5)


def _stack_store(case, ctx, view=0, max_layers=5):
    store = R.OracleStore()
    layers = case["layers"] + [case["layers"][0]]
    seeds = [case["seed"] + i for i in range(4)] + [case["seed"] + 9]
    for i, ((sl, Cl), sd) in enumerate(zip(layers[:max_layers], seeds)):
        w = attention_weights(Cl, case["ctx_dim"], sd)
        x = seeded((1, sl * sl, Cl), case["seed"] + 100 + 10 * view + i)
        R.hooked_attention(x, ctx, *w, case["heads"], store, case["R"])
    return store


def test_g2_collect_maps(golden):
    g = golden("g2_collect_maps.npz")
    c = STACK_CASE
    ctx = seeded((1, c["T"], c["ctx_dim"]), c["seed"] + 200)
    idx = torch.tensor(c["indices"])
    cases = {
        "res-1": dict(upsample_res=-1),
        "resR": dict(upsample_res=c["R"]),
        "res24": dict(upsample_res=24),
        "res-1_idx": dict(upsample_res=-1, indices=idx),
        "res40_idx": dict(upsample_res=40, indices=idx),
        "res-1_layers02": dict(upsample_res=-1, layers=[0, 2]),
    }
    for tag, kw in cases.items():
        store = _stack_store(c, ctx)
        assert len(store.step_store["attn"]) == 4          # the 5th layer is gated out
        m = R.collect_maps(store, **kw)
        assert len(store.step_store["attn"]) == 0          # reset (optimize.py:77)
        torch.testing.assert_close(m, t(g[tag]), **TOL)
    # invariant: probabilities over tokens sum to 1 at every pixel
    m = R.collect_maps(_stack_store(c, ctx), upsample_res=-1)
    torch.testing.assert_close(m.sum(0), torch.ones(c["R"], c["R"]), rtol=1e-5, atol=1e-5)


def test_g3_selection(golden):
    g = golden("g3_selection.npz")
    s = SEL_CASE
    maps, maps_t = selection_maps()
    assert torch.equal(R.find_max_pixel(maps), t(g["find_max_pixel"]))
    assert R.find_max_pixel(maps)[5].tolist() == [3.5, 4.5]          # first index wins the tie
    assert torch.equal(R.find_k_max_pixels(maps, 1), t(g["find_k_max_pixels_1"]))
    assert torch.equal(R.find_k_max_pixels(maps, 2), t(g["find_k_max_pixels_2"]))
    assert torch.equal(R.mask_radius(maps[:3], R.find_max_pixel(maps[:3]), 0.05 * s["R"]), t(g["mask_radius"]))
    for ns in (1, 2):
        kl = R.gaussian_kl(maps, s["sigma"], num_subjects=ns)
        torch.testing.assert_close(kl, t(g[f"kl_ns{ns}"]), rtol=1e-5, atol=1e-6)
        top = R.find_top_k_gaussian(maps, s["n_cand"], sigma=s["sigma"], num_subjects=ns)
        assert torch.equal(top, t(g[f"top_k_gaussian_ns{ns}"]))
        fps = R.furthest_point_sampling(maps_t, s["top_k"], top)
        assert torch.equal(fps, t(g[f"fps_ns{ns}"]))


def test_g3b_entropy_strategy(golden):
    """top_k_strategy == 'entropy' (ptp_utils.py:165-187, optimize.py:382-385)."""
    g = golden("g3b_entropy.npz")
    s = SEL_CASE
    maps, maps_t = selection_maps()
    torch.testing.assert_close(R.token_entropy(maps), t(g["entropy"]), rtol=1e-6, atol=1e-6)
    sm = sharp_entropy_maps(maps)
    torch.testing.assert_close(R.token_entropy(sm), t(g["entropy_sharp"]), rtol=1e-6, atol=1e-6)
    assert torch.equal(R.entropy_sort(sm, s["n_cand"]), t(g["entropy_sort_sharp"]))
    top = R.entropy_sort(maps, s["n_cand"])
    assert torch.equal(top, t(g["entropy_sort"]))
    assert torch.equal(R.furthest_point_sampling(maps_t, s["top_k"], top), t(g["fps_entropy"]))


def test_g9_weighted_average_keypoints(golden):
    """eval.pixel_from_weighted_avg (eval.py:113-155) on the reference's own augmented-inference maps."""
    g = golden("g9_reference_augmented_tiny.npz")
    maps = t(g["maps"])
    size = float(maps.shape[-1])
    torch.testing.assert_close(R.pixel_from_weighted_avg(maps.clone()) / size, t(g["keypoints_weighted"]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(R.pixel_from_weighted_avg(maps.clone(), distance=3), t(g["weighted_d3"]), rtol=1e-6, atol=1e-5)
    torch.testing.assert_close(R.pixel_from_weighted_avg(maps.clone(), distance=-1), t(g["weighted_all"]), rtol=1e-6, atol=1e-5)
    m = maps.clone()
    R.pixel_from_weighted_avg(m)
    assert (m == 0).sum() > (maps == 0).sum()                 # in-place zeroing, like the reference


def test_g10_full_size_launch_shape(golden):
    """BASELINE config 2's real shape (SD-1.5 hooked layers, R = 128, T = 77): oracle vs the reference's checksums."""
    from oracle.fixtures import FULL_CASE as fc
    g = golden("g10_full_size.npz")
    ctx = seeded((1, fc["T"], fc["ctx_dim"]), fc["seed"] + 200)
    store = R.OracleStore()
    with torch.no_grad():
        for i, (sl, Cl) in enumerate(fc["layers"]):
            w = attention_weights(Cl, fc["ctx_dim"], fc["seed"] + i)
            R.hooked_attention(seeded((1, sl * sl, Cl), fc["seed"] + 100 + i), ctx, *w, fc["heads"], store, fc["R"])
        m = R.collect_maps(store, upsample_res=-1)
    flat = m.reshape(fc["T"], -1)
    assert torch.equal(flat.argmax(dim=-1), t(g["argmax"]))
    torch.testing.assert_close(flat.double().sum(dim=-1), t(g["token_sum"]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(m.reshape(-1)[:: fc["stride"]], t(g["strided"]), rtol=1e-4, atol=1e-7)
    assert abs(m.double().sum().item() - float(g["checksum"])) < 1e-3          # == R*R = 16384 up to rounding
    assert abs(float(g["checksum"]) - fc["R"] ** 2) < 1e-2


def test_g4_losses(golden):
    g = golden("g4_losses.npz")
    s = SEL_CASE
    maps, maps_t = selection_maps()
    sel = torch.tensor(s["sel"])
    thetas = t(g["theta"])
    torch.testing.assert_close(torch.cat([R.affine_matrix(11.0, 0.87, (0.13, -0.21)),
                                          R.affine_matrix(-14.0, 0.95, (-0.2, 0.05))]), thetas, rtol=0, atol=0)
    for ns in (1, 2):
        a = maps[sel].clone().requires_grad_(True)
        l = R.sharpening_loss(a, sigma=s["sigma"], num_subjects=ns)
        l.backward()
        assert abs(l.item() - float(g[f"sharp_ns{ns}"])) <= 1e-6 * abs(float(g[f"sharp_ns{ns}"]))
        torch.testing.assert_close(a.grad, t(g[f"sharp_grad_ns{ns}"]), **TOL)
    for index in (0, 1):
        a = maps[sel].clone().requires_grad_(True)
        b = maps_t[sel].clone().requires_grad_(True)
        l = R.equivariance_loss(a, b, thetas, index)
        l.backward()
        assert abs(l.item() - float(g[f"equiv_{index}"])) <= 1e-6 * abs(float(g[f"equiv_{index}"]))
        torch.testing.assert_close(a.grad, t(g[f"equiv_grad_a_{index}"]), **TOL)
        torch.testing.assert_close(b.grad, t(g[f"equiv_grad_b_{index}"]), **TOL)


def test_g7_gauss_affine(golden):
    g = golden("g7_gauss_affine.npz")
    pos = torch.tensor([[[0.3, 0.7], [0.5, 0.5], [0.02, 0.98]], [[0.9, 0.1], [0.25, 0.75], [0.6, 0.4]]])
    torch.testing.assert_close(R.gaussian_circle(pos[0], 24, 2.0), t(g["gaussian_circle"]), **TOL)
    torch.testing.assert_close(R.gaussian_circles(pos, 24, 3.0), t(g["gaussian_circles"]), **TOL)
    thetas = torch.cat([R.affine_matrix(11.0, 0.87, (0.13, -0.21)), R.affine_matrix(-14.0, 0.95, (-0.2, 0.05))])
    img = seeded((2, 3, 20, 28), 77).abs()
    w = R.affine_warp(img, thetas)
    torch.testing.assert_close(w, t(g["warp"]), **TOL)
    torch.testing.assert_close(R.affine_unwarp(w, thetas), t(g["unwarp"]), **TOL)
    # random draw order (invertable_transform.py:42-51): 4 uniforms per image from the global RNG
    torch.manual_seed(1234)
    th = []
    for _ in range(2):
        a, sc, tr = R.draw_affine_params(lambda: torch.rand(1).item(), 15, (0.8, 1.0), (0.25, 0.25))
        th.append(R.affine_matrix(a, sc, tr))
    torch.testing.assert_close(torch.cat(th), t(g["theta_seed1234"]), rtol=0, atol=0)


def test_g5_subgraph_gradient_and_adam(golden):
    g = golden("g5_subgraph.npz")
    e = E2E_CASE
    context = seeded((1, e["T"], e["ctx_dim"]), e["seed"] + 200).requires_grad_(True)
    opt = torch.optim.Adam([context], lr=5e-3)
    maps = []
    for view in (0, 1):
        store = _stack_store(e, context, view=view, max_layers=4)
        maps.append(R.collect_maps(store, upsample_res=-1, layers=[0, 1, 2, 3]))
    am, am_t = maps
    torch.testing.assert_close(am, t(g["map"]), **TOL)
    torch.testing.assert_close(am_t, t(g["map_t"]), **TOL)
    theta = R.affine_matrix(11.0, 0.87, (0.13, -0.21))
    loss, sharp, equiv, sel = R.image_loss(am, am_t, theta, 0, furthest_point_num_samples=e["n_cand"],
                                           top_k=e["top_k"], sigma=e["sigma"])
    assert torch.equal(sel, t(g["sel"]))
    assert abs(sharp.item() - float(g["sharp"])) < 1e-5 * abs(float(g["sharp"]))
    assert abs(equiv.item() - float(g["equiv"])) < 1e-5 * abs(float(g["equiv"]))
    loss.backward()
    torch.testing.assert_close(context.grad, t(g["context_grad"]), rtol=1e-4, atol=1e-7)
    opt.step()
    torch.testing.assert_close(context.detach(), t(g["context_after_adam"]), rtol=1e-6, atol=1e-6)


def test_scheduler_restatement():
    ts = R.ddim_timesteps()
    assert ts[-1].item() == 0 and ts[0].item() == 980 and len(ts) == 50
    acp = R.ddim_alphas_cumprod()
    assert abs(acp[0].item() - (1 - 0.00085)) < 1e-7


def test_g8_oracle_full_step_vs_reference_driver(golden):
    """The oracle's CPU step (oracle/cpu_path.py) against the REFERENCE's own run_and_find_attn / selection /
    losses / backward, both driving the same reduced-width SD-topology module tree (G8)."""
    from oracle import cpu_path
    from oracle.fixtures import TINY_CASE as tc
    from stablekeypoints_amd.optimize_token import load_ldm
    g = golden("g8_reference_step_tiny.npz")
    ldm, _, _ = load_ldm("cpu", "tiny", feature_upsample_res=tc["R"])
    store = R.OracleStore()
    cpu_path.register_reference_hook(ldm.unet, store, tc["R"])
    image = torch.rand(1, 3, tc["size"], tc["size"], generator=torch.Generator().manual_seed(tc["seed"]))
    ctx = seeded((1, tc["T"], 768), tc["seed"] + 1).requires_grad_(True)
    noise = t(g["noise"])
    loss, sharp, equiv, sel, am, am_t = cpu_path.image_step(
        ldm, image, ctx, store, t(g["theta"]), noise[0:1], noise[1:2],
        furthest_point_num_samples=tc["n_cand"], top_k=tc["top_k"], sigma=tc["sigma"])
    torch.testing.assert_close(am, t(g["map"]), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(am_t, t(g["map_t"]), rtol=1e-4, atol=1e-7)
    assert torch.equal(sel, t(g["sel"]))
    assert abs(sharp.item() - float(g["sharp"])) < 1e-5 * abs(float(g["sharp"]))
    assert abs(equiv.item() - float(g["equiv"])) < 1e-4 * abs(float(g["equiv"]))
    loss.backward()
    ref = t(g["context_grad"])
    torch.testing.assert_close(ctx.grad, ref, rtol=1e-3, atol=1e-5 * ref.abs().max().item())


def _loop_inputs():
    from oracle.fixtures import LOOP_CASE as lc
    images = torch.rand(lc["n_images"], 3, lc["size"], lc["size"], generator=torch.Generator().manual_seed(lc["seed"]))
    ctx0 = seeded((1, lc["T"], 768), lc["seed"] + 1) * lc["ctx_gain"]
    return lc, images, ctx0


def test_g11_oracle_trajectory_vs_reference_optimize_embedding(golden):
    """oracle/cpu_path.optimize_trajectory against the REFERENCE's own `optimize.optimize_embedding` loop
    (optimize.py:269-452: 3 optimizer steps x 2 accumulated images, Adam) on the reduced-width model, fed the loader
    order / noise / thetas the reference drew (G11): the embedding after every optimizer step."""
    from oracle import cpu_path
    from stablekeypoints_amd.optimize_token import load_ldm
    g = golden("g11_reference_trajectory_tiny.npz")
    lc, images, ctx0 = _loop_inputs()
    ldm, _, _ = load_ldm("cpu", "tiny", feature_upsample_res=lc["R"])
    traj = cpu_path.optimize_trajectory(ldm, images, ctx0, g["order"], t(g["noise"]), t(g["thetas"]), steps=lc["steps"],
                                        accum=lc["accum"], R_up=lc["R"], furthest_point_num_samples=lc["n_cand"],
                                        top_k=lc["top_k"], sigma=lc["sigma"])
    ref = t(g["context"])
    assert ref.shape == (lc["steps"], lc["T"], 768)
    torch.testing.assert_close(traj, ref, rtol=1e-5, atol=1e-6)
    # every step moved the embedding by about lr (Adam's first steps are sign-like), cumulatively
    assert 0.9 * 5e-3 < (ref[0] - ctx0[0]).abs().max().item() < 1.1 * 5e-3
    assert (ref[2] - ctx0[0]).abs().max().item() > 2.5 * 5e-3


def test_g12_oracle_best_indices_vs_reference_find_best_indices(golden):
    """oracle/cpu_path.find_best_indices against the REFERENCE's own `keypoint_regressor.find_best_indices` (:16-108)
    over 24 images (G12): per-image selections and the voted indices, exact."""
    from oracle import cpu_path
    from stablekeypoints_amd.optimize_token import load_ldm
    g11, g = golden("g11_reference_trajectory_tiny.npz"), golden("g12_reference_best_indices_tiny.npz")
    lc, images, _ = _loop_inputs()
    ldm, _, _ = load_ldm("cpu", "tiny", feature_upsample_res=lc["R"])
    out, picked = cpu_path.find_best_indices(ldm, images, t(g11["context"])[-1][None], g["order"], t(g["noise"]),
                                             R_up=lc["R"], furthest_point_num_samples=lc["n_cand"], top_k=lc["top_k"],
                                             sigma=lc["sigma"])
    assert len(g["order"]) == lc["num_indices"] == 24
    assert torch.equal(picked, t(g["per_image"]))
    assert torch.equal(out, t(g["indices"]))
