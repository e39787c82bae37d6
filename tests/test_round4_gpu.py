"""GPU parity added in round 4.

* the drop-in boundary for main.py's stages 2-5: the REFERENCE-ORDER `collect_maps` / `find_best_indices`
  restatements of oracle/ applied to the store `load_ldm` returns BY DEFAULT (FusedAttn handles, tensor duck type);
* G13: the reference's own `keypoint_regressor.precompute_all_keypoints` (dataset loop -> augmented inference -> final
  keypoint locations) against the product's batched dataset-level driver;
* G11 replayed on TWO ranks (the reference's accumulation of 2 images == 2 ranks x 1 image + one SUM all-reduce);
* a whole step at BASELINE config 2's launch shape (full-width SD-1.5, 512^2, 4 images x 2 views) against the oracle.
"""
import os
import sys

import numpy as np
import pytest
import torch
from _tol import assert_grad_close

from oracle import ref_path as R
from oracle.fixtures import KPTS_CASE as kc, LOOP_CASE as lc, seeded

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def t(a):
    return torch.from_numpy(np.asarray(a))


class _Images(torch.utils.data.Dataset):
    def __init__(self, data, kpts=None):
        self.data, self.kpts = data, kpts

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        return {"img": self.data[i]} if self.kpts is None else {"img": self.data[i], "kpts": self.kpts[i]}


def _loop_images():
    return torch.rand(lc["n_images"], 3, lc["size"], lc["size"], generator=torch.Generator().manual_seed(lc["seed"]))


def _loop_args(**over):
    from stablekeypoints_amd.optimize import default_args
    kw = dict(num_tokens=lc["T"], feature_upsample_res=lc["R"], furthest_point_num_samples=lc["n_cand"],
              top_k=lc["top_k"], sigma=lc["sigma"], batch_size=lc["accum"], num_steps=lc["steps"],
              image_size=lc["size"], device="cuda", log_interval=0, num_indices=lc["num_indices"])
    kw.update(over)
    return default_args(**kw)


# ---------------------------------------------------------------------------------------------------------------------
# boundary: reference-order code on the DEFAULT store
# ---------------------------------------------------------------------------------------------------------------------
def test_reference_order_collect_maps_on_default_store(golden):
    """main.py stages 2-5 keep calling the reference's `ptp_utils.run_and_find_attn` -> `optimize.collect_maps`
    (optimize.py:44-77: reshape / index / permute / interpolate / stack+mean over the stored entries) on the controller
    `load_ldm` returns.  Here that op sequence (oracle.ref_path.collect_maps) runs on the product's DEFAULT store --
    `FusedAttn` handles on the GPU -- after a full, un-exited forward like the reference's `find_pred_noise`, and must
    give the fused kernel's maps (rtol 1e-3) for every `upsample_res` / `indices` form the callers use; then the
    reference-order `find_best_indices` loop (keypoint_regressor.py:56-108 restated with oracle functions on those maps)
    must vote the product's -- and G12's -- indices."""
    from stablekeypoints_amd import ptp_utils
    from stablekeypoints_amd._maps import FusedAttn, collect_maps
    from stablekeypoints_amd.keypoint_regressor import find_best_indices
    from stablekeypoints_amd.optimize_token import load_ldm
    assert torch.cuda.is_available()
    ldm, controllers, n = load_ldm("cuda", "tiny", feature_upsample_res=lc["R"])
    dev, ctrl = next(iter(controllers.items()))
    assert ctrl.materialize is False
    g11, g12 = golden("g11_reference_trajectory_tiny.npz"), golden("g12_reference_best_indices_tiny.npz")
    images, ctx = _loop_images(), t(g11["context"])[-1][None].cuda()
    noise = t(g12["noise"])
    idx = torch.tensor([5, 0, 11, 5])
    with torch.no_grad():
        for kw in (dict(upsample_res=-1), dict(upsample_res=lc["R"]), dict(upsample_res=48, indices=idx),
                   dict(upsample_res=-1, layers=(0, 2)), dict(upsample_res=-1, indices=idx)):
            got = []
            for fn in (R.collect_maps, collect_maps):
                ptp_utils.find_pred_noise(ldm, images[:2].cuda(), ctx, device=dev, noise=noise[:2].cuda(), early_exit=False,
                                          controllers=controllers)
                store = ctrl.step_store["attn"]
                assert len(store) == 4 and all(isinstance(e, FusedAttn) for e in store)
                assert tuple(store[0].shape) == (2 * store[0].heads, lc["R"] ** 2, lc["T"])
                got.append(fn(ctrl, **kw))
                assert len(ctrl.step_store["attn"]) == 0                # both reset the controller
            assert got[0].shape == got[1].shape and got[0].is_cuda
            torch.testing.assert_close(got[0], got[1], rtol=1e-3, atol=1e-6)
        # stage 2 in the reference's order: one image per forward, reference collect_maps on the handles, python selection
        picked = []
        for it, i in enumerate(g12["order"]):
            ptp_utils.find_pred_noise(ldm, images[int(i)][None].cuda(), ctx, device=dev, noise=noise[it:it + 1].cuda(),
                                      early_exit=False, controllers=controllers)
            am = R.collect_maps(ctrl, upsample_res=lc["R"], layers=(0, 1, 2, 3)).cpu()
            cand = R.find_top_k_gaussian(am, lc["n_cand"], sigma=lc["sigma"], num_subjects=1)
            picked.append(R.furthest_point_sampling(am, lc["top_k"], cand))
        flat = torch.cat(picked)
        ids, counts = torch.unique(flat, return_counts=True)
        voted = ids[counts.argsort(descending=True)][:lc["top_k"]]
    ds = _Images(images.cuda())
    from stablekeypoints_amd import keypoint_regressor
    real = keypoint_regressor.build_dataset
    keypoint_regressor.build_dataset = lambda args: ds
    try:
        votes = []
        prod = find_best_indices(ldm, ctx, _loop_args(), controllers, n, draws=(g12["order"], noise), votes_out=votes)
    finally:
        keypoint_regressor.build_dataset = real
    assert torch.equal(torch.stack(picked), votes[0]), "reference-order loop on the default store != fused find_best_indices"
    assert torch.equal(voted, prod) and torch.equal(voted, t(g12["indices"]))


# ---------------------------------------------------------------------------------------------------------------------
# G13: dataset-level keypoint driver
# ---------------------------------------------------------------------------------------------------------------------
def _kpts_dataset():
    gen = torch.Generator().manual_seed(kc["seed"])
    data = torch.rand(kc["n_images"], 3, kc["size"], kc["size"], generator=gen)
    kpts = torch.rand(kc["n_images"], kc["n_kpts"], 2, generator=gen)
    return _Images(data, kpts)


def _kpts_args(strategy, group):
    return _loop_args(augmentation_iterations=kc["aug_iters"], max_num_points=50_000, max_loc_strategy=strategy,
                      images_per_forward=group)


@pytest.mark.parametrize("group", [2, 5, 1])
def test_g13_precompute_all_keypoints_vs_reference(golden, group):
    """The product's `precompute_all_keypoints` (several images' views per network batch, fused un-warp/accumulate) fed the
    loader order, noise and thetas of the REFERENCE's own `keypoint_regressor.precompute_all_keypoints` run over a
    5-image keypoint dataset (G13; the embedding of G11, the indices of G12): FINAL KEYPOINT LOCATIONS -- arg-max
    strategy bit-exact (a location is an integer pixel + 0.5 over 512), weighted-average strategy rtol 1e-3; targets pass
    through in loader order."""
    from stablekeypoints_amd.keypoint_regressor import precompute_all_keypoints
    from stablekeypoints_amd.optimize_token import load_ldm
    g11, g = golden("g11_reference_trajectory_tiny.npz"), golden("g13_reference_keypoints_tiny.npz")
    ldm, controllers, n = load_ldm("cuda", "tiny", feature_upsample_res=lc["R"])
    ctx = t(g11["context"])[-1][None].cuda()
    draws = (g["order"], t(g["noise"]), t(g["thetas"]))
    for strategy, key in (("argmax", "source_argmax"), ("weighted_avg", "source_weighted")):
        src, tgt, vis = precompute_all_keypoints(ldm, ctx, t(g["indices"]), _kpts_args(strategy, group), controllers, n,
                                                 dataset=_kpts_dataset(), draws=draws)
        assert vis is None and src.shape == (kc["n_images"], lc["top_k"], 2) and torch.equal(tgt, t(g["target"]))
        ref = t(g[key])
        print(f"G13 {strategy} (group {group}): max |diff| {(src.cpu() - ref).abs().max().item():.3e}")
        if strategy == "argmax":
            assert torch.equal(src.cpu(), ref)
        else:
            torch.testing.assert_close(src.cpu(), ref, rtol=1e-3, atol=1e-6)
    # no annotations (synthetic / custom datasets): locations only
    src2, tgt2, vis2 = precompute_all_keypoints(ldm, ctx, t(g["indices"]), _kpts_args("argmax", group), controllers, n,
                                                dataset=_Images(_kpts_dataset().data), draws=draws)
    assert tgt2 is None and vis2 is None and torch.equal(src2.cpu(), t(g["source_argmax"]))


# ---------------------------------------------------------------------------------------------------------------------
# G11 on two ranks
# ---------------------------------------------------------------------------------------------------------------------
def _g11_rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from stablekeypoints_amd import dist as D, keypoint_regressor, optimize
    from stablekeypoints_amd.optimize_token import load_ldm
    D.init_from_env("gloo")                                        # both ranks on cuda:0; gloo stands in for RCCL
    torch.cuda.set_device(0)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g11_reference_trajectory_tiny.npz")))
    images = _loop_images()
    ds = _Images(images.cuda())
    optimize.build_dataset = keypoint_regressor.build_dataset = lambda args: ds
    ldm, controllers, n = load_ldm("cuda", "tiny", feature_upsample_res=lc["R"])
    ctx0 = seeded((1, lc["T"], 768), lc["seed"] + 1) * lc["ctx_gain"]
    if rank == 1:
        ctx0 = ctx0 + 1.0                                           # the reducer must broadcast rank 0's start embedding
    # the reference's iteration j = step * accum + i becomes rank i of step `step`: order / thetas j, noise rows 2j, 2j+1
    its = list(range(rank, lc["steps"] * lc["accum"], world))
    noise = t(g["noise"])
    draws = (g["order"][its], torch.cat([noise[2 * j:2 * j + 2] for j in its]), t(g["thetas"])[its])
    traj = []
    out = optimize.optimize_embedding(ldm, _loop_args(batch_size=lc["accum"]), controllers, n, context=ctx0.clone(),
                                      draws=draws, trajectory_out=traj)
    assert torch.equal(out[0], traj[-1][0])
    # stage 2 + 3 on the two ranks: images sharded, votes / locations gathered
    g12 = dict(np.load(os.path.join(ROOT, "tests", "golden", "g12_reference_best_indices_tiny.npz")))
    per = lc["num_indices"] // world
    mine = list(range(rank, lc["num_indices"], world))[:per]
    ctx = t(g["context"])[-1][None].cuda()
    idx = keypoint_regressor.find_best_indices(ldm, ctx, _loop_args(), controllers, n,
                                               draws=(g12["order"][mine], t(g12["noise"])[mine]))
    g13 = dict(np.load(os.path.join(ROOT, "tests", "golden", "g13_reference_keypoints_tiny.npz")))
    src, tgt, _ = keypoint_regressor.precompute_all_keypoints(
        ldm, ctx, t(g13["indices"]), _kpts_args("argmax", 2), controllers, n, dataset=_kpts_dataset(),
        draws=(g13["order"], t(g13["noise"]), t(g13["thetas"])))
    torch.save({"traj": torch.cat(traj).cpu(), "idx": idx.cpu(), "src": src.cpu(), "tgt": tgt.cpu()},
               os.path.join(out_dir, f"r{rank}.pt"))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_g11_two_ranks_vs_reference(tmp_path, golden):
    """a13 / (e) pinned to the reference's OWN loop: G11 is `optimize_embedding` with 2 accumulated images per optimizer
    step (optimize.py:405-425: mean over the mini-batch, `/ accum`, backward per image, step every `accum`).  Replayed as
    2 ranks x 1 image with the draws sliced `rank::2`, one SUM all-reduce of the embedding gradient per step and the same
    Adam on both ranks: the embedding after each of the 3 steps must meet the single-rank G11 tolerances AND be
    bit-identical across the ranks.  The same two ranks then run `find_best_indices` (votes all-gathered) and
    `precompute_all_keypoints` (locations all-gathered) against G12 / G13."""
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    port = 29700 + (os.getpid() % 200)
    mp.spawn(_g11_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["traj"], b["traj"]), "the embedding diverged between the ranks"
    g = golden("g11_reference_trajectory_tiny.npz")
    ref, got = t(g["context"]), a["traj"]
    ctx0 = (seeded((1, lc["T"], 768), lc["seed"] + 1) * lc["ctx_gain"])[0]
    lr = 5e-3
    assert got.shape == ref.shape
    for s in range(lc["steps"]):                                 # same bar as test_g11_optimize_embedding_trajectory_vs_reference (observed: 0.035 lr)
        err = ((got[s] - ctx0) - (ref[s] - ctx0)).abs()
        print(f"2 ranks, step {s + 1}: displacement error mean {err.mean().item() / lr:.5f} lr, max {err.max().item() / lr:.3f} lr")
        assert err.mean().item() < 1e-3 * lr and err.max().item() < 0.12 * lr
    assert torch.equal(a["idx"], b["idx"]) and torch.equal(a["idx"], t(golden("g12_reference_best_indices_tiny.npz")["indices"]))
    g13 = golden("g13_reference_keypoints_tiny.npz")
    assert torch.equal(a["src"], b["src"]) and torch.equal(a["src"], t(g13["source_argmax"])) and torch.equal(a["tgt"], t(g13["target"]))


# ---------------------------------------------------------------------------------------------------------------------
# a whole step at config 2's launch shape
# ---------------------------------------------------------------------------------------------------------------------
_ORACLE_IMAGE_STEPS = {}          # image index -> the oracle's step for that image (test_sd15_config2_shape_step_vs_oracle)


@pytest.mark.parametrize("n_img", [4, 1])
def test_sd15_config2_shape_step_vs_oracle(n_img, sd15_cpu):
    """n_img = 1 is BASELINE config 3's PER-RANK shape (8 images over 8 ranks: 1 image x 2 views, B = 2 rows -- the K-split
    plans, persistent-walk grids and map launches all change with the row count); n_img = 4:
    BASELINE config 2's LAUNCH SHAPE -- full-width SD-1.5, 512^2, 4 images x 2 views (B = 8 rows), T = 77, R = 128, K = 10 of
    25 -- one fused `group_step` on the MI355X (GroupNorm folded into the 128 -> 128 @512^2 Winograd convolution, N = 4096
    flash attention, the persistent conv walk, the B = 8 map launch) against the oracle's reference-order CPU step run per
    image (`oracle/cpu_path.image_step`).  Maps rtol 1e-3 (north_star), selected tokens exact (or the stated near-tie
    rule), losses 1e-3 / 2e-3, embedding gradient rtol 5e-3 against the SUM of the per-image oracle gradients."""
    from oracle import cpu_path
    from stablekeypoints_amd import ptp_utils
    from stablekeypoints_amd._maps import collect_maps_batched
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import default_args, group_step, image_losses
    from stablekeypoints_amd.optimize_token import load_ldm
    assert torch.cuda.is_available()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    Rup, T, n_cand, top_k = 128, 77, 25, 10
    ldm, controllers, _ = load_ldm("cuda", "sd15", feature_upsample_res=Rup)
    cpu = sd15_cpu
    # one set of inputs for both cases (four images; the 1-image case is image 0 with its two noise rows): the oracle's per-image
    # step depends on (image, embedding, theta, the two noise rows) only, so its result for an image is computed once per session
    g = torch.Generator().manual_seed(7)
    images4 = torch.rand(4, 3, 512, 512, generator=g)
    ctx = torch.randn(1, T, 768, generator=g) * 5.0
    noise4 = torch.randn(8, 4, 64, 64, generator=g)                 # rows 0..3: the images, 4..7: their affine copies
    thetas4 = torch.cat([R.affine_matrix(a, s, tr) for a, s, tr in
                         ((9.0, 0.9, (0.1, -0.15)), (-12.0, 0.85, (-0.2, 0.05)), (4.0, 0.97, (0.0, 0.22)), (-7.0, 0.8, (0.18, 0.1)))])
    images, thetas = images4[:n_img], thetas4[:n_img]
    noise = torch.cat([noise4[:n_img], noise4[4:4 + n_img]])
    args = default_args(num_tokens=T, feature_upsample_res=Rup, furthest_point_num_samples=n_cand, top_k=top_k, batch_size=n_img)
    store = R.OracleStore()
    assert cpu_path.register_reference_hook(cpu.unet, store, Rup) == 18
    ref_maps, ref_sel, ref_sharp, ref_equiv = [], [], 0.0, 0.0
    gref = torch.zeros(1, T, 768)
    for i in range(n_img):
        if i not in _ORACLE_IMAGE_STEPS:
            c_ref = ctx.clone().requires_grad_(True)
            loss, sharp, equiv, sel, am, am_t = cpu_path.image_step(
                cpu, images[i:i + 1], c_ref, store, thetas[i:i + 1], noise[i:i + 1], noise[n_img + i:n_img + i + 1],
                furthest_point_num_samples=n_cand, top_k=top_k, sigma=args.sigma)
            loss.backward()
            _ORACLE_IMAGE_STEPS[i] = (c_ref.grad.clone(), am.detach(), am_t.detach(), sel, sharp.item(), equiv.item())
        g_i, am, am_t, sel, sharp_i, equiv_i = _ORACLE_IMAGE_STEPS[i]
        gref += g_i / n_img                                          # optimize.py:418-420: loss / accumulation steps
        ref_maps.append((am, am_t))
        ref_sel.append(sel)
        ref_sharp += sharp_i / n_img
        ref_equiv += equiv_i / n_img
    dev, controller = next(iter(controllers.items()))
    tr = RandomAffineWithInverse()
    with torch.no_grad():
        both = torch.cat([images.cuda(), tr(images.cuda(), theta=thetas)])
        ptp_utils.find_pred_noise(ldm, both, ctx.cuda(), device=dev, noise=noise.cuda(), early_exit=True, controllers=controllers)
        assert [tuple(r.q.shape) for r in controller.step_store["attn"]] == [(2 * n_img, 256, 1280)] * 3 + [(2 * n_img, 1024, 640)]
        maps = collect_maps_batched(controller)
    worst, ties = 0.0, 0
    for i in range(n_img):
        am, am_t = ref_maps[i]
        worst = max(worst, ((maps[i].cpu() - am).abs() / am.abs().clamp_min(1e-6)).max().item())
        torch.testing.assert_close(maps[i].cpu(), am, rtol=1e-3, atol=1e-6)
        torch.testing.assert_close(maps[n_img + i].cpu(), am_t, rtol=1e-3, atol=1e-6)
        _, _, sel_g = image_losses(maps[i], maps[n_img + i], thetas[i].reshape(-1).tolist(), args)
        if not torch.equal(sel_g.cpu(), ref_sel[i]):
            s, _ = R.gaussian_kl(am, args.sigma).sort()
            gaps = (s[1:n_cand + 1] - s[:n_cand]) / s[:n_cand].abs()
            assert gaps.min().item() < 1e-4, f"image {i}: selection differs although every score gap is decisive"
            ties += 1
    print("config-2 shape maps: max rel diff", worst, "near-tie images", ties)
    assert ties == 0, "a KL near-tie changed a selection: losses / gradient not comparable on this seed"
    c_gpu = ctx.clone().cuda().requires_grad_(True)
    loss_g, eq_g, sh_g = group_step(ldm, images, c_gpu, args, controller, tr, denom=n_img, noise=noise.cuda(), thetas=thetas)
    assert abs(sh_g.item() - ref_sharp) < 1e-3 * abs(ref_sharp)
    assert abs(eq_g.item() - ref_equiv) < 2e-3 * abs(ref_equiv)
    print("config-2 shape grad: |g|max", gref.abs().max().item(), "max abs diff", (c_gpu.grad.cpu() - gref).abs().max().item())
    assert_grad_close(c_gpu.grad, gref, "test_round4_gpu.py#1")
    if n_img == 1:
        # config 3's per-rank step is launch-bound on the host and runs as a captured hipGraph (optimize.GraphedStep, the product's
        # rule for groups of <= 2 images): the replayed step against the same oracle gradient
        from stablekeypoints_amd.optimize import GraphedStep
        c_rep = ctx.clone().cuda().requires_grad_(True)
        graphed = GraphedStep(ldm, c_rep, args, controller, tr, denom=n_img, warmup=0)
        lg = graphed(images, noise=noise.cuda(), thetas=thetas)
        assert graphed.state[1] != "eager" and graphed.state[1]["graph"] is not None
        assert abs(lg[2].item() - ref_sharp) < 1e-3 * abs(ref_sharp) and abs(lg[1].item() - ref_equiv) < 2e-3 * abs(ref_equiv)
        assert_grad_close(c_rep.grad, gref, "config 3 per-rank step, replayed from the hipGraph")


# ---------------------------------------------------------------------------------------------------------------------
# map backward without the dV staging (column sweep)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [
    dict(sides=[16, 16, 16, 32], H=8, T=77, R=128, B=2, K=10),            # BASELINE config 2 launch shape: quads + pairs
    dict(sides=[16, 32], H=8, T=128, R=128, B=1, K=16),                    # the kernel's largest T and K
    dict(sides=[32], H=3, T=16, R=128, B=3, K=2),                          # one natural chunk, odd head count, pairs only
    dict(sides=[32, 64], H=2, T=40, R=256, B=1, K=5),                      # R = 256: one column half per workgroup
    dict(sides=[16], H=4, T=50, R=128, B=2, K=10),                         # T % 16 != 0: pad tokens in the last chunk
    dict(sides=[16, 32], H=8, T=300, R=128, B=1, K=10),                    # a wide token axis (the wide forward gives lse)
])
def test_column_sweep_map_backward_vs_fp64_dense_and_sweep(case, monkeypatch, route="col"):
    """skp_attn_map_bwd_col_f32 (column sweep: vertical adjoint in a register window, horizontal adjoint on complete low-res
    rows, no dV staging; the T <= 128 route of the step since round 4) against fp64 autograd through
    F.interpolate(bicubic) + softmax, against the dense-gradient kernels and against the token-major sweep on the same
    inputs; repeat run bit-identical; pad columns written as 0."""
    from test_round3_gpu import _fp64_map_and_grad
    from stablekeypoints_amd import ops
    sides, H, T, R, B, K = (case[k] for k in ("sides", "H", "T", "R", "B", "K"))
    assert ops.map_bwd_col_supported(sides, K, R, T, H)
    monkeypatch.setattr(ops, "COL_MAX_T", 1024)
    g = torch.Generator().manual_seed(17)
    NT = (T + 15) // 16 * 16
    S = []
    for s in sides:
        S_l = torch.zeros(B, H, s * s, NT)
        S_l[..., :T] = torch.randn(B, H, s * s, T, generator=g) * 3.0
        S.append(S_l.cuda())
    sel = torch.stack([torch.randperm(T, generator=g)[:K] for _ in range(B)]).cuda()
    G = torch.randn(B, K, R, R, generator=g).cuda()
    M, lse = ops._map_fwd(S, sides, B, H, T, R)
    _, dz = _fp64_map_and_grad(S, sides, H, T, R, sel, G)
    monkeypatch.setattr(ops, "MAP_BWD_MODE", route)
    dS = ops._map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse)
    dS2 = ops._map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse)
    dM = torch.zeros(B, T, R, R, device="cuda")
    for b in range(B):
        dM[b, sel[b]] = G[b]
    dD = [torch.zeros_like(s_) for s_ in S]
    ops._map_bwd(S, dD, sides, B, H, T, R, dM, lse)
    dW = None
    if max(sides) <= ops.MAP_SPARSE_MAX_SIDE:
        monkeypatch.setattr(ops, "MAP_BWD_MODE", "sweep")
        dW = ops._map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse)
    for l in range(len(sides)):
        ref = dz[l]
        scale = ref.abs().max().item()
        assert torch.equal(dS[l], dS2[l])                                     # deterministic
        assert torch.isfinite(dS[l]).all() and (dS[l][..., T:] == 0).all()    # pad columns written as 0
        err = (dS[l][..., :T].double() - ref).abs().max().item()
        err_dense = (dD[l][..., :T].double() - ref).abs().max().item()
        print(f"layer {l} (s={sides[l]}): |dz|max {scale:.3e}  {route} err {err / scale:.2e}  dense err {err_dense / scale:.2e}")
        assert err < 2e-5 * scale
        torch.testing.assert_close(dS[l][..., :T], dD[l][..., :T], rtol=1e-4, atol=2e-5 * scale)
        if dW is not None:
            torch.testing.assert_close(dS[l][..., :T], dW[l][..., :T], rtol=1e-4, atol=2e-5 * scale)

