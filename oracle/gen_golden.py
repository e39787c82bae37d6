"""Generate tests/golden/*.npz by running the REFERENCE's own functions (TEST INFRASTRUCTURE).

Runs only in the build container (needs /root/reference).  Recipe = SURVEY.md section 8(c):
third-party modules that are absent offline (diffusers, wandb, h5py, torchvision, cv2) are
MagicMock'ed, the reference's `datasets/` namespace dir is pre-registered so the installed
HuggingFace `datasets` does not shadow it, then `optimize_token` is imported first.

The fixtures hold DATA only: seeds/shapes of the inputs (inputs are regenerated with
`oracle.fixtures.seeded`), and the reference's outputs.  Nothing of the reference's source
text is stored.

    python -m oracle.gen_golden          # rewrites tests/golden/
"""
from __future__ import annotations

import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    sys.path.insert(0, REF)
    for m in ["diffusers", "wandb", "h5py", "torchvision", "torchvision.transforms",
              "torchvision.transforms.functional", "cv2"]:
        sys.modules[m] = MagicMock()
    ds = types.ModuleType("datasets")
    ds.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = ds
    from unsupervised_keypoints import optimize_token  # noqa: F401  (import order matters)
    from unsupervised_keypoints import ptp_utils, optimize, eval as ref_eval, invertable_transform
    return optimize_token, ptp_utils, optimize, ref_eval, invertable_transform


# --- a stand-in module tree with the diffusers==0.8.0 CrossAttention attribute layout ----
class CrossAttention(torch.nn.Module):          # class name is what the reference matches on
    def __init__(self, query_dim, context_dim, heads):
        super().__init__()
        self.heads = heads
        self.scale = (query_dim // heads) ** -0.5
        self.to_q = torch.nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = torch.nn.Linear(context_dim, query_dim, bias=False)
        self.to_v = torch.nn.Linear(context_dim, query_dim, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(query_dim, query_dim), torch.nn.Dropout(0.0)])

    def reshape_heads_to_batch_dim(self, t):
        b, n, c = t.shape
        h = self.heads
        return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def reshape_batch_dim_to_heads(self, t):
        bh, n, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, d * h)


class Net(torch.nn.Module):
    def __init__(self, mods):
        super().__init__()
        self.up_blocks = torch.nn.ModuleList(mods)


def main():
    from oracle.fixtures import seeded, HOOK_CASES, STACK_CASE, load_weights_into, SEL_CASE, E2E_CASE
    optimize_token, ptp_utils, optimize, ref_eval, inv = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    # ---------------- G1: the hook on one module -------------------------------------
    g1 = {}
    for name, c in HOOK_CASES.items():
        mod = CrossAttention(c["C"], c["ctx_dim"], c["heads"])
        load_weights_into(mod, c["seed"])
        mod_self = CrossAttention(c["C"], c["C"], c["heads"])      # attn1-style (context=None)
        load_weights_into(mod_self, c["seed"] + 1000)
        net = Net([mod, mod_self])
        ctrl = ptp_utils.AttentionStore()
        ptp_utils.register_attention_control(net, ctrl, feature_upsample_res=c["R"])
        x = seeded((c["B"], c["s"] * c["s"], c["C"]), c["seed"] + 100)
        ctx = seeded((c["B"], c["T"], c["ctx_dim"]), c["seed"] + 200)
        with torch.no_grad():
            out = mod.forward(x, context=ctx)
        p_up = ctrl.step_store["attn"][0]
        assert ctrl.num_att_layers == 2 and len(ctrl.step_store["attn"]) == 1
        if c["full"]:
            g1[name + "/p_up"] = p_up.numpy()
            g1[name + "/out"] = out.numpy()
        else:  # SD-shape case: strided sample + sums
            g1[name + "/p_up_strided"] = p_up.reshape(-1)[:: c["stride"]].numpy()
            g1[name + "/out_strided"] = out.reshape(-1)[:: c["stride"]].numpy()
            g1[name + "/p_up_sum"] = np.array(p_up.double().sum().item())
        # self-attention call (context=None) must not store
        with torch.no_grad():
            out_self = mod_self.forward(x)
        assert len(ctrl.step_store["attn"]) == 1
        g1[name + "/out_self_strided"] = out_self.reshape(-1)[::7].numpy()
    np.savez_compressed(os.path.join(OUT, "g1_hook.npz"), **g1)

    # ---------------- G2: collect_maps on a 4-layer store -----------------------------
    c = STACK_CASE
    mods = [CrossAttention(Cl, c["ctx_dim"], c["heads"]) for (sl, Cl) in c["layers"]]
    for i, m in enumerate(mods):
        load_weights_into(m, c["seed"] + i)
    # a 5th qualifying cross layer that must NOT be stored (gate len<4, ptp_utils.py:511)
    extra = CrossAttention(c["layers"][0][1], c["ctx_dim"], c["heads"])
    load_weights_into(extra, c["seed"] + 9)
    net = Net(mods + [extra])
    ctrl = ptp_utils.AttentionStore()
    ptp_utils.register_attention_control(net, ctrl, feature_upsample_res=c["R"])
    assert ctrl.num_att_layers == 5
    ctx = seeded((1, c["T"], c["ctx_dim"]), c["seed"] + 200)

    def run_stack():
        for i, (m, (sl, Cl)) in enumerate(zip(mods + [extra], c["layers"] + [c["layers"][0]])):
            x = seeded((1, sl * sl, Cl), c["seed"] + 100 + i)
            m.forward(x, context=ctx)
        assert len(ctrl.step_store["attn"]) == 4

    g2 = {}
    idx = torch.tensor(c["indices"])
    with torch.no_grad():
        for tag, kw in {
            "res-1": dict(upsample_res=-1),
            "resR": dict(upsample_res=c["R"]),
            "res24": dict(upsample_res=24),
            "res-1_idx": dict(upsample_res=-1, indices=idx),
            "res40_idx": dict(upsample_res=40, indices=idx),
            "res-1_layers02": dict(upsample_res=-1, layers=[0, 2]),
        }.items():
            run_stack()
            m = optimize.collect_maps(ctrl, **kw)
            assert len(ctrl.step_store["attn"]) == 0        # collect_maps resets
            g2[tag] = m.numpy()
    np.savez_compressed(os.path.join(OUT, "g2_collect_maps.npz"), **g2)

    # ---------------- G3: selection on fixed maps -------------------------------------
    s = SEL_CASE
    from oracle.fixtures import selection_maps
    maps, maps_t = selection_maps()
    g3 = {}
    g3["find_max_pixel"] = ref_eval.find_max_pixel(maps).numpy()
    g3["find_k_max_pixels_1"] = ref_eval.find_k_max_pixels(maps, num=1).numpy()
    g3["find_k_max_pixels_2"] = ref_eval.find_k_max_pixels(maps, num=2).numpy()
    g3["mask_radius"] = ref_eval.mask_radius(maps[:3], ref_eval.find_max_pixel(maps[:3]), 0.05 * maps.shape[1]).numpy()
    for ns in (1, 2):
        top = ptp_utils.find_top_k_gaussian(maps, s["n_cand"], sigma=s["sigma"], num_subjects=ns)
        g3[f"top_k_gaussian_ns{ns}"] = top.numpy()
        # also the KL values themselves (restated from the reference lines, via its helpers)
        loc = ref_eval.find_k_max_pixels(maps, num=ns) / maps.shape[1]
        sm = torch.softmax(maps.view(maps.shape[0], -1) + 1e-5, dim=-1)
        tgt = optimize_token.gaussian_circles(loc, size=maps.shape[1], sigma=s["sigma"], device="cpu")
        tgt = tgt.reshape(maps.shape[0], -1) + 1e-5
        tgt = tgt / tgt.sum(dim=-1, keepdim=True)
        g3[f"kl_ns{ns}"] = torch.sum(tgt * (torch.log(tgt) - torch.log(sm)), dim=-1).numpy()
        fps = ptp_utils.furthest_point_sampling(maps_t, s["top_k"], top)
        g3[f"fps_ns{ns}"] = fps.numpy()
    np.savez_compressed(os.path.join(OUT, "g3_selection.npz"), **g3)

    # ---------------- G3b: the 'entropy' strategy (ptp_utils.entropy_sort) -------------
    g3b = {}
    ent_top = ptp_utils.entropy_sort(maps, s["n_cand"])
    g3b["entropy_sort"] = ent_top.numpy()
    sm = torch.softmax(maps.view(maps.shape[0], -1), dim=-1)
    g3b["entropy"] = torch.distributions.Categorical(probs=sm).entropy().numpy()
    g3b["fps_entropy"] = ptp_utils.furthest_point_sampling(maps_t, s["top_k"], ent_top).numpy()
    # a sharper family of maps (entropies far apart) so the ranking is not decided by rounding
    from oracle.fixtures import sharp_entropy_maps
    sharp_maps = sharp_entropy_maps(maps)
    g3b["entropy_sort_sharp"] = ptp_utils.entropy_sort(sharp_maps, s["n_cand"]).numpy()
    g3b["entropy_sharp"] = torch.distributions.Categorical(
        probs=torch.softmax(sharp_maps.view(maps.shape[0], -1), dim=-1)).entropy().numpy()
    np.savez_compressed(os.path.join(OUT, "g3b_entropy.npz"), **g3b)

    # ---------------- G4: losses (+ gradients) and G7: gaussian / affine ---------------
    g4 = {}
    theta = inv.RandomAffineWithInverse().create_affine_matrix(11.0, 0.87, (0.13, -0.21))
    theta2 = inv.RandomAffineWithInverse().create_affine_matrix(-14.0, 0.95, (-0.2, 0.05))
    thetas = torch.cat([theta, theta2], dim=0)
    g4["theta"] = thetas.numpy()
    sel = torch.tensor(s["sel"])
    for ns in (1, 2):
        a = maps[sel].clone().requires_grad_(True)
        l = optimize.sharpening_loss(a, sigma=s["sigma"], device="cpu", num_subjects=ns)
        l.backward()
        g4[f"sharp_ns{ns}"] = np.array(l.item(), dtype=np.float64)
        g4[f"sharp_grad_ns{ns}"] = a.grad.numpy()
    tr = inv.RandomAffineWithInverse()
    tr.last_params = {"theta": thetas}
    for index in (0, 1):
        a = maps[sel].clone().requires_grad_(True)
        b = maps_t[sel].clone().requires_grad_(True)
        l = optimize.equivariance_loss(a, b[None].repeat(2, 1, 1, 1), tr, index)
        l.backward()
        g4[f"equiv_{index}"] = np.array(l.item(), dtype=np.float64)
        g4[f"equiv_grad_a_{index}"] = a.grad.numpy()
        g4[f"equiv_grad_b_{index}"] = b.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g4_losses.npz"), **g4)

    g7 = {}
    pos = torch.tensor([[[0.3, 0.7], [0.5, 0.5], [0.02, 0.98]], [[0.9, 0.1], [0.25, 0.75], [0.6, 0.4]]])
    g7["gaussian_circle"] = optimize_token.gaussian_circle(pos[0], size=24, sigma=2.0, device="cpu").numpy()
    g7["gaussian_circles"] = optimize_token.gaussian_circles(pos, size=24, sigma=3.0, device="cpu").numpy()
    img = seeded((2, 3, 20, 28), 77).abs()
    tr = inv.RandomAffineWithInverse(degrees=15, scale=(0.8, 1.0), translate=(0.25, 0.25))
    g7["warp"] = tr(img, theta=thetas).numpy()
    g7["unwarp"] = tr.inverse(tr(img, theta=thetas)).numpy()
    # the random draw order: seed torch's global RNG, record theta
    torch.manual_seed(1234)
    _ = tr(img)
    g7["theta_seed1234"] = tr.last_params["theta"].numpy()
    np.savez_compressed(os.path.join(OUT, "g7_gauss_affine.npz"), **g7)

    # ---------------- G5/G6: hook-subgraph gradient and one Adam step ------------------
    e = E2E_CASE
    mods = [CrossAttention(Cl, e["ctx_dim"], e["heads"]) for (sl, Cl) in e["layers"]]
    for i, m in enumerate(mods):
        load_weights_into(m, e["seed"] + i)
        for p in m.parameters():
            p.requires_grad = False
    net = Net(mods)
    ctrl = ptp_utils.AttentionStore()
    ptp_utils.register_attention_control(net, ctrl, feature_upsample_res=e["R"])
    context = seeded((1, e["T"], e["ctx_dim"]), e["seed"] + 200).requires_grad_(True)
    opt = torch.optim.Adam([context], lr=5e-3)

    def forward_maps(view):
        for i, (m, (sl, Cl)) in enumerate(zip(mods, e["layers"])):
            x = seeded((1, sl * sl, Cl), e["seed"] + 100 + 10 * view + i)
            m.forward(x, context=context)
        return optimize.collect_maps(ctrl, upsample_res=-1, layers=[0, 1, 2, 3])

    am = forward_maps(0)
    am_t = forward_maps(1)
    top = ptp_utils.find_top_k_gaussian(am, e["n_cand"], sigma=e["sigma"], num_subjects=1)
    sel = ptp_utils.furthest_point_sampling(am_t, e["top_k"], top)
    tr = inv.RandomAffineWithInverse()
    tr.last_params = {"theta": theta}
    sharp = optimize.sharpening_loss(am[sel], device="cpu", sigma=e["sigma"], num_subjects=1)
    equiv = optimize.equivariance_loss(am[sel], am_t[sel][None].repeat(1, 1, 1, 1), tr, 0)
    loss = equiv * 1000.0 + sharp * 100.0
    loss.backward()
    g5 = {
        "map": am.detach().numpy(), "map_t": am_t.detach().numpy(), "cand": top.numpy(),
        "sel": sel.numpy(), "sharp": np.array(sharp.item()), "equiv": np.array(equiv.item()),
        "loss": np.array(loss.item()), "context_grad": context.grad.numpy().copy(),
    }
    opt.step()
    g5["context_after_adam"] = context.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g5_subgraph.npz"), **g5)

    # ---------------- G8 / G9: the REFERENCE's own driver functions on the build's reduced-width SD-topology
    # model (CPU): run_and_find_attn -> selection -> losses -> backward (G8) and
    # run_image_with_context_augmented (G9).  The random draws (noise, thetas) are recorded so the
    # replacement can be fed the same ones.
    from oracle.fixtures import TINY_CASE
    from stablekeypoints_amd.ldm.pipeline import StableDiffusionPipeline
    from stablekeypoints_amd.ldm.scheduler import DDIMScheduler
    tc = TINY_CASE
    sch = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                        set_alpha_to_one=False)
    sch.set_timesteps(50)
    ldm = StableDiffusionPipeline.from_pretrained("tiny", scheduler=sch)
    for pm in list(ldm.unet.parameters()) + list(ldm.vae.parameters()):
        pm.requires_grad = False
    ctrl = ptp_utils.AttentionStore()
    controllers = {torch.device("cpu"): ctrl}
    ptp_utils.register_attention_control(ldm.unet, ctrl, feature_upsample_res=tc["R"])     # the reference's hook
    assert ctrl.num_att_layers == 18
    image = torch.rand(1, 3, tc["size"], tc["size"], generator=torch.Generator().manual_seed(tc["seed"]))
    context = seeded((1, tc["T"], 768), tc["seed"] + 1).requires_grad_(True)
    drawn = []
    real_randn_like = torch.randn_like

    def rec_randn_like(x, *a, **k):
        out = real_randn_like(x, *a, **k)
        drawn.append(out.clone())
        return out

    torch.randn_like = rec_randn_like
    try:
        torch.manual_seed(tc["seed"] + 2)
        kw = dict(layers=[0, 1, 2, 3], noise_level=-1, from_where=["down_cross", "mid_cross", "up_cross"],
                  upsample_res=-1, device="cpu", controllers=controllers)
        am = ptp_utils.run_and_find_attn(ldm, image, context, **kw)[0]
        tr = inv.RandomAffineWithInverse(degrees=15, scale=(0.8, 1.0), translate=(0.25, 0.25))
        timg = tr(image)
        theta8 = tr.last_params["theta"].clone()
        am_t = ptp_utils.run_and_find_attn(ldm, timg, context, **kw)[0]
        top = ptp_utils.find_top_k_gaussian(am, tc["n_cand"], sigma=tc["sigma"], num_subjects=1)
        sel = ptp_utils.furthest_point_sampling(am_t, tc["top_k"], top)
        sharp = optimize.sharpening_loss(am[sel], device="cpu", sigma=tc["sigma"], num_subjects=1)
        equiv = optimize.equivariance_loss(am[sel], am_t[sel][None].repeat(1, 1, 1, 1), tr, 0)
        loss = equiv * 1000.0 + sharp * 100.0
        loss.backward()
        g8 = {"noise": torch.cat(drawn).numpy(), "theta": theta8.numpy(), "map": am.detach().numpy(),
              "map_t": am_t.detach().numpy(), "cand": top.numpy(), "sel": sel.numpy(),
              "sharp": np.array(sharp.item()), "equiv": np.array(equiv.item()),
              "context_grad": context.grad.numpy().copy()}
        np.savez_compressed(os.path.join(OUT, "g8_reference_step_tiny.npz"), **g8)

        # G9: augmented inference with the reference's eval.run_image_with_context_augmented
        drawn.clear()
        thetas9 = []
        real_call = inv.RandomAffineWithInverse.__call__

        def rec_call(self, img, theta=None):
            out = real_call(self, img, theta)
            thetas9.append(self.last_params["theta"].clone())
            return out

        inv.RandomAffineWithInverse.__call__ = rec_call
        torch.manual_seed(tc["seed"] + 3)
        with torch.no_grad():
            idx = sel.detach().clone()
            maps9 = ref_eval.run_image_with_context_augmented(
                ldm, image[0].permute(1, 2, 0).numpy(), context.detach(), idx, device="cpu",
                from_where=["down_cross", "mid_cross", "up_cross"], layers=[0, 1, 2, 3],
                augmentation_iterations=tc["aug_iters"], noise_level=-1, augment_degrees=30, augment_scale=(0.9, 1.1),
                augment_translate=(0.1, 0.1), controllers=controllers, num_gpus=1, upscale_size=tc["upscale"])
        inv.RandomAffineWithInverse.__call__ = real_call
        kp = ref_eval.find_max_pixel(maps9) / float(tc["upscale"])
        # max_loc_strategy == "weighted_avg" (keypoint_regressor.py:191-196 -> eval.pixel_from_weighted_avg); the
        # reference zeroes far pixels of its argument in place, so it gets a copy
        kp_w = ref_eval.pixel_from_weighted_avg(maps9.clone()) / float(tc["upscale"])
        kp_w3 = ref_eval.pixel_from_weighted_avg(maps9.clone(), distance=3)
        kp_wall = ref_eval.pixel_from_weighted_avg(maps9.clone(), distance=-1)
        g9 = {"noise": torch.cat(drawn).numpy(), "thetas": torch.cat(thetas9).numpy(), "indices": idx.numpy(),
              "maps": maps9.numpy(), "keypoints": kp.numpy(), "keypoints_weighted": kp_w.numpy(),
              "weighted_d3": kp_w3.numpy(), "weighted_all": kp_wall.numpy()}
        np.savez_compressed(os.path.join(OUT, "g9_reference_augmented_tiny.npz"), **g9)
    finally:
        torch.randn_like = real_randn_like

    # ---------------- G10: full-size launch shape of BASELINE config 2 ------------------
    from oracle.fixtures import FULL_CASE
    fc = FULL_CASE
    mods = [CrossAttention(Cl, fc["ctx_dim"], fc["heads"]) for (sl, Cl) in fc["layers"]]
    for i, m in enumerate(mods):
        load_weights_into(m, fc["seed"] + i)
    net = Net(mods)
    ctrl = ptp_utils.AttentionStore()
    ptp_utils.register_attention_control(net, ctrl, feature_upsample_res=fc["R"])
    ctx = seeded((1, fc["T"], fc["ctx_dim"]), fc["seed"] + 200)
    with torch.no_grad():
        for i, (m, (sl, Cl)) in enumerate(zip(mods, fc["layers"])):
            m.forward(seeded((1, sl * sl, Cl), fc["seed"] + 100 + i), context=ctx)
        assert len(ctrl.step_store["attn"]) == 4 and tuple(ctrl.step_store["attn"][0].shape) == (8, 128 * 128, 77)
        mfull = optimize.collect_maps(ctrl, upsample_res=-1, layers=[0, 1, 2, 3])
    assert tuple(mfull.shape) == (fc["T"], fc["R"], fc["R"])
    flat = mfull.reshape(fc["T"], -1)
    g10 = {"argmax": flat.argmax(dim=-1).numpy(), "token_sum": flat.double().sum(dim=-1).numpy(),
           "token_max": flat.max(dim=-1).values.numpy(), "strided": mfull.reshape(-1)[:: fc["stride"]].numpy(),
           "checksum": np.array(mfull.double().sum().item()),
           "weighted_checksum": np.array((mfull.double().reshape(-1) * torch.linspace(0.5, 1.5, mfull.numel(), dtype=torch.float64)).sum().item())}
    np.savez_compressed(os.path.join(OUT, "g10_full_size.npz"), **g10)

    # ---------------- G11 / G12: the reference's OWN loops -- optimize.optimize_embedding (optimize.py:269-452) for three
    # optimizer steps of two accumulated images, and keypoint_regressor.find_best_indices (:16-108) over 24 images (three
    # groups of eight in the replacement) -- on the reduced-width SD-topology model.  The dataset is a seeded tensor stub
    # patched in for `custom_images.CustomDataset`; the loader order, every noise draw and every theta are recorded so
    # the replacement can be fed the same ones.  The reference hard-codes `.to('cuda:0')` for the two scalar losses
    # (optimize.py:405-406): on this GPU-less box `Tensor.to` maps that one argument to 'cpu' during the run.
    g11, g12 = reference_loops(optimize, ptp_utils, inv)
    np.savez_compressed(os.path.join(OUT, "g11_reference_trajectory_tiny.npz"), **g11)
    np.savez_compressed(os.path.join(OUT, "g12_reference_best_indices_tiny.npz"), **g12)

    # ---------------- G13: the reference's OWN keypoint_regressor.precompute_all_keypoints (:111-198) ----------------------
    g13 = reference_keypoints(ptp_utils, inv, g11, g12)
    np.savez_compressed(os.path.join(OUT, "g13_reference_keypoints_tiny.npz"), **g13)

    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden written to", OUT, "total bytes", tot)


def reference_loops(optimize, ptp_utils, inv):
    """Drive the reference's `optimize_embedding` and `find_best_indices` on CPU -> (g11, g12) fixture dicts."""
    import types as _types
    from oracle.fixtures import LOOP_CASE, seeded
    from stablekeypoints_amd.ldm.pipeline import StableDiffusionPipeline
    from stablekeypoints_amd.ldm.scheduler import DDIMScheduler
    from unsupervised_keypoints import keypoint_regressor
    from datasets import custom_images
    lc = LOOP_CASE
    sch = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                        set_alpha_to_one=False)
    sch.set_timesteps(50)
    ldm = StableDiffusionPipeline.from_pretrained("tiny", scheduler=sch)
    for pm in list(ldm.unet.parameters()) + list(ldm.vae.parameters()):
        pm.requires_grad = False
    ctrl = ptp_utils.AttentionStore()
    controllers = {torch.device("cpu"): ctrl}
    ptp_utils.register_attention_control(ldm.unet, ctrl, feature_upsample_res=lc["R"])

    served = []

    class TensorImages(torch.utils.data.Dataset):
        def __init__(self, data_root=None, image_size=512):
            self.data = torch.rand(lc["n_images"], 3, lc["size"], lc["size"],
                                   generator=torch.Generator().manual_seed(lc["seed"]))

        def __len__(self):
            return self.data.shape[0]

        def __getitem__(self, i):
            served.append(int(i))
            return {"img": self.data[i]}

    drawn, thetas, after_step = [], [], []
    real_randn_like, real_call, real_to = torch.randn_like, inv.RandomAffineWithInverse.__call__, torch.Tensor.to
    real_adam_step, real_dataset = torch.optim.Adam.step, custom_images.CustomDataset

    def rec_randn_like(x, *a, **k):
        out = real_randn_like(x, *a, **k)
        drawn.append(out.clone())
        return out

    def rec_call(self, img, theta=None):
        out = real_call(self, img, theta)
        thetas.append(self.last_params["theta"].clone())
        return out

    def to_cpu(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            a = ("cpu",) + a[1:]
        return real_to(self, *a, **k)

    def rec_adam_step(self, *a, **k):
        out = real_adam_step(self, *a, **k)
        after_step.append(self.param_groups[0]["params"][0].detach().clone())
        return out

    args = _types.SimpleNamespace(
        dataset_name="custom", dataset_loc="", max_len=-1, device="cpu", lr=5e-3, num_steps=lc["steps"],
        num_tokens=lc["T"], batch_size=lc["accum"], top_k_strategy="gaussian", feature_upsample_res=lc["R"],
        furthest_point_num_samples=lc["n_cand"], top_k=lc["top_k"], num_subjects=1, sharpening_loss_weight=100,
        equivariance_attn_loss_weight=1000, layers=[0, 1, 2, 3], noise_level=-1, sigma=lc["sigma"],
        augment_degrees=15, augment_scale=[0.8, 1.0], augment_translate=[0.25, 0.25], wandb=False,
        num_indices=lc["num_indices"])
    ctx0 = seeded((1, lc["T"], 768), lc["seed"] + 1) * lc["ctx_gain"]
    torch.randn_like, inv.RandomAffineWithInverse.__call__ = rec_randn_like, rec_call
    torch.Tensor.to, torch.optim.Adam.step, custom_images.CustomDataset = to_cpu, rec_adam_step, TensorImages
    try:
        torch.manual_seed(lc["seed"] + 2)
        final = optimize.optimize_embedding(ldm, args, controllers, 1, context=ctx0.clone())
        assert len(after_step) == lc["steps"] and torch.equal(final, after_step[-1])
        assert len(served) == lc["steps"] * lc["accum"] and len(drawn) == 2 * len(served) == 2 * len(thetas)
        g11 = {"order": np.array(served), "noise": torch.cat(drawn).numpy(), "thetas": torch.cat(thetas).numpy(),
               "context": torch.cat(after_step).numpy()}
        served.clear(); drawn.clear(); thetas.clear()

        per_image = []
        real_fps = ptp_utils.furthest_point_sampling

        def rec_fps(*a, **k):
            out = real_fps(*a, **k)
            per_image.append(out.clone())
            return out

        ptp_utils.furthest_point_sampling = rec_fps
        torch.manual_seed(lc["seed"] + 3)
        best = keypoint_regressor.find_best_indices(ldm, final, args, controllers, 1)
        ptp_utils.furthest_point_sampling = real_fps
        assert len(served) == lc["num_indices"] == len(drawn) == len(per_image) and not thetas
        g12 = {"order": np.array(served), "noise": torch.cat(drawn).numpy(),
               "per_image": torch.stack(per_image).numpy(), "indices": best.numpy()}
    finally:
        torch.randn_like, inv.RandomAffineWithInverse.__call__ = real_randn_like, real_call
        torch.Tensor.to, torch.optim.Adam.step, custom_images.CustomDataset = real_to, real_adam_step, real_dataset
    return g11, g12


def reference_keypoints(ptp_utils, inv, g11, g12, hook="reference", controllers=None, ldm=None):
    """Drive the reference's `precompute_all_keypoints` (dataset loop -> `run_image_with_context_augmented` ->
    `find_max_pixel / 512` or `pixel_from_weighted_avg / 512`) on CPU over a keypoint-dataset stub patched in for
    `taichi.TrainRegSet`, once per `max_loc_strategy` from the same seed (same loader order, noise and thetas), recording
    every draw.  `hook="reference"`: the reference's own hook on the reduced-width model (the golden).  With `ldm` /
    `controllers` given (oracle/check_dropin.py) the same call runs on THOSE objects instead."""
    import types as _types
    from oracle.fixtures import LOOP_CASE, KPTS_CASE
    from unsupervised_keypoints import keypoint_regressor
    lc, kc = LOOP_CASE, KPTS_CASE
    if ldm is None:
        from stablekeypoints_amd.ldm.pipeline import StableDiffusionPipeline
        from stablekeypoints_amd.ldm.scheduler import DDIMScheduler
        sch = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                            set_alpha_to_one=False)
        sch.set_timesteps(50)
        ldm = StableDiffusionPipeline.from_pretrained("tiny", scheduler=sch)
        for pm in list(ldm.unet.parameters()) + list(ldm.vae.parameters()):
            pm.requires_grad = False
        ctrl = ptp_utils.AttentionStore()
        controllers = {torch.device("cpu"): ctrl}
        ptp_utils.register_attention_control(ldm.unet, ctrl, feature_upsample_res=lc["R"])
    served = []

    class KeypointImages(torch.utils.data.Dataset):
        def __init__(self, data_root=None, image_size=512, **_):
            gen = torch.Generator().manual_seed(kc["seed"])
            self.data = torch.rand(kc["n_images"], 3, kc["size"], kc["size"], generator=gen)
            self.kpts = torch.rand(kc["n_images"], kc["n_kpts"], 2, generator=gen)

        def __len__(self):
            return self.data.shape[0]

        def __getitem__(self, i):
            served.append(int(i))
            return {"img": self.data[i], "kpts": self.kpts[i]}

    drawn, thetas = [], []
    real_randn_like, real_call = torch.randn_like, inv.RandomAffineWithInverse.__call__
    real_set = keypoint_regressor.taichi.TrainRegSet

    def rec_randn_like(x, *a, **k):
        out = real_randn_like(x, *a, **k)
        drawn.append(out.clone())
        return out

    def rec_call(self, img, theta=None):
        out = real_call(self, img, theta)
        thetas.append(self.last_params["theta"].clone())
        return out

    context = torch.from_numpy(g11["context"][-1:]).clone()
    indices = torch.from_numpy(g12["indices"]).clone()
    out = {"indices": indices.numpy()}
    torch.randn_like, inv.RandomAffineWithInverse.__call__ = rec_randn_like, rec_call
    keypoint_regressor.taichi.TrainRegSet = KeypointImages
    try:
        for strategy in ("argmax", "weighted_avg"):
            args = _types.SimpleNamespace(
                dataset_name="taichi", dataset_loc="", device="cpu", layers=[0, 1, 2, 3], noise_level=-1,
                augmentation_iterations=kc["aug_iters"], augment_degrees=15, augment_scale=[0.8, 1.0],
                augment_translate=[0.25, 0.25], save_folder="outputs", max_num_points=50_000, max_loc_strategy=strategy)
            served.clear(); drawn.clear(); thetas.clear()
            torch.manual_seed(kc["seed"] + 1)
            src, tgt, vis = keypoint_regressor.precompute_all_keypoints(ldm, context, indices, args, controllers, 1)
            assert vis is None and len(served) == kc["n_images"] and len(drawn) == len(thetas) == kc["n_images"] * kc["aug_iters"]
            if strategy == "argmax":
                out.update(order=np.array(served), noise=torch.cat(drawn).numpy(), thetas=torch.cat(thetas).numpy(),
                           source_argmax=src.numpy(), target=tgt.numpy())
            else:
                assert np.array_equal(out["order"], np.array(served)) and np.array_equal(out["thetas"], torch.cat(thetas).numpy())
                out["source_weighted"] = src.numpy()
    finally:
        torch.randn_like, inv.RandomAffineWithInverse.__call__ = real_randn_like, real_call
        keypoint_regressor.taichi.TrainRegSet = real_set
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g13":          # only the newest fixture (the others are unchanged inputs to it)
        _o = import_reference()
        g11_ = dict(np.load(os.path.join(OUT, "g11_reference_trajectory_tiny.npz")))
        g12_ = dict(np.load(os.path.join(OUT, "g12_reference_best_indices_tiny.npz")))
        np.savez_compressed(os.path.join(OUT, "g13_reference_keypoints_tiny.npz"), **reference_keypoints(_o[1], _o[4], g11_, g12_))
    else:
        main()
