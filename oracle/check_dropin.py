"""INTEGRATION.md section A, checked in the build container (TEST INFRASTRUCTURE; needs /root/reference).

`unsupervised_keypoints/main.py` with the two imports of section A re-bound keeps calling the REFERENCE's own
`keypoint_regressor.find_best_indices` (main.py:221-227), `keypoint_regressor.precompute_all_keypoints` (:250-257)
and `eval.evaluate` (:307-315); all three reach the reference's `ptp_utils.run_and_find_attn` ->
`optimize.collect_maps` (optimize.py:27-79), which reshapes / indexes / permutes / interpolates / stacks the entries
of `controller.step_store["attn"]`.  The build's `load_ldm` stores `FusedAttn` handles there; this script shows that
the reference's functions run UNCHANGED on them (the handle's tensor duck type) and reproduce what the reference
computes with its own hook:

  stage 2  reference find_best_indices on load_ldm("cpu", "tiny")'s controllers  == golden G12 (per-image picks, vote)
  stage 3  reference precompute_all_keypoints on the same objects                  == golden G13 (source keypoints)
  stage 5  reference eval.run_image_with_context_augmented (the body of evaluate)  == golden G9  (maps, keypoints)

On the host the module tree is driven on CPU tensors (no HIP kernels can run in this container); the same handles on
an MI355X are covered by tests/test_round4_gpu.py::test_reference_order_collect_maps_on_default_store.

    python -m oracle.check_dropin
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from oracle.gen_golden import import_reference, reference_keypoints, OUT


def main():
    from oracle.fixtures import LOOP_CASE, TINY_CASE, seeded
    optimize_token, ptp_utils, optimize, ref_eval, inv = import_reference()
    from unsupervised_keypoints import keypoint_regressor
    from datasets import custom_images
    from stablekeypoints_amd.optimize_token import load_ldm            # line 1 of INTEGRATION.md A's two-line change
    from stablekeypoints_amd._maps import FusedAttn
    torch.set_num_threads(8)
    lc = LOOP_CASE
    g9 = dict(np.load(os.path.join(OUT, "g9_reference_augmented_tiny.npz")))
    g11 = dict(np.load(os.path.join(OUT, "g11_reference_trajectory_tiny.npz")))
    g12 = dict(np.load(os.path.join(OUT, "g12_reference_best_indices_tiny.npz")))
    g13 = dict(np.load(os.path.join(OUT, "g13_reference_keypoints_tiny.npz")))

    ldm, controllers, num_gpus = load_ldm("cpu", "tiny", feature_upsample_res=lc["R"])
    ctrl = next(iter(controllers.values()))
    assert num_gpus == 1 and not ctrl.materialize                       # the DEFAULT store: handles, not tensors

    # the stored entries really are handles when the reference's collect_maps sees them
    seen = []
    real_collect = optimize.collect_maps

    def spy_collect(controller, *a, **k):
        seen.extend(type(e) for e in controller.step_store["attn"])
        return real_collect(controller, *a, **k)

    ptp_utils.collect_maps = spy_collect                                # the name ptp_utils.run_and_find_attn resolves

    # ---- stage 2: main.py:221-227 ------------------------------------------------------------------------------
    class TensorImages(torch.utils.data.Dataset):
        def __init__(self, data_root=None, image_size=512):
            self.data = torch.rand(lc["n_images"], 3, lc["size"], lc["size"],
                                   generator=torch.Generator().manual_seed(lc["seed"]))

        def __len__(self):
            return self.data.shape[0]

        def __getitem__(self, i):
            return {"img": self.data[i]}

    args = types.SimpleNamespace(
        dataset_name="custom", dataset_loc="", device="cpu", top_k_strategy="gaussian", feature_upsample_res=lc["R"],
        furthest_point_num_samples=lc["n_cand"], top_k=lc["top_k"], num_subjects=1, layers=[0, 1, 2, 3], noise_level=-1,
        sigma=lc["sigma"], num_indices=lc["num_indices"])
    per_image = []
    real_fps, real_set = ptp_utils.furthest_point_sampling, custom_images.CustomDataset

    def rec_fps(*a, **k):
        out = real_fps(*a, **k)
        per_image.append(out.clone())
        return out

    ptp_utils.furthest_point_sampling, custom_images.CustomDataset = rec_fps, TensorImages
    try:
        final = torch.from_numpy(g11["context"][-1:]).clone()
        torch.manual_seed(lc["seed"] + 3)                               # G12's seed: same loader order, same noise
        best = keypoint_regressor.find_best_indices(ldm, final, args, controllers, num_gpus)
    finally:
        ptp_utils.furthest_point_sampling, custom_images.CustomDataset = real_fps, real_set
    assert seen and all(t is FusedAttn for t in seen), seen
    same = int((torch.stack(per_image).numpy() == g12["per_image"]).all(axis=1).sum())
    print(f"stage 2  find_best_indices (reference code, build's load_ldm): ran over {len(per_image)} images; "
          f"{same}/{len(per_image)} per-image selections identical to G12; vote {best.tolist()} vs {g12['indices'].tolist()}")
    assert np.array_equal(best.numpy(), g12["indices"]) and same == len(per_image)

    # ---- stage 3: main.py:250-257 ------------------------------------------------------------------------------
    got = reference_keypoints(ptp_utils, inv, g11, g12, controllers=controllers, ldm=ldm)
    for key in ("source_argmax", "source_weighted"):
        d = np.abs(got[key] - g13[key]).max()
        print(f"stage 3  precompute_all_keypoints ({key}): max |diff| vs G13 = {d:.3e}")
    assert np.array_equal(got["order"], g13["order"]) and np.array_equal(got["source_argmax"], g13["source_argmax"])
    np.testing.assert_allclose(got["source_weighted"], g13["source_weighted"], rtol=0, atol=2e-5)

    # ---- stage 5: main.py:307-315 -> eval.evaluate's per-image call (eval.py:428-446) -------------------------------
    tc = TINY_CASE
    ldm9, controllers9, _ = load_ldm("cpu", "tiny", feature_upsample_res=tc["R"])
    image = torch.rand(1, 3, tc["size"], tc["size"], generator=torch.Generator().manual_seed(tc["seed"]))
    context = seeded((1, tc["T"], 768), tc["seed"] + 1)
    torch.manual_seed(tc["seed"] + 3)
    with torch.no_grad():
        maps9 = ref_eval.run_image_with_context_augmented(
            ldm9, image[0].permute(1, 2, 0).numpy(), context, torch.from_numpy(g9["indices"]), device="cpu",
            from_where=["down_cross", "mid_cross", "up_cross"], layers=[0, 1, 2, 3], augmentation_iterations=tc["aug_iters"],
            noise_level=-1, augment_degrees=30, augment_scale=(0.9, 1.1), augment_translate=(0.1, 0.1),
            controllers=controllers9, num_gpus=1, upscale_size=tc["upscale"])
    kp = ref_eval.find_max_pixel(maps9) / float(tc["upscale"])
    rel = float((maps9 - torch.from_numpy(g9["maps"])).abs().max() / torch.from_numpy(g9["maps"]).abs().max())
    print(f"stage 5  run_image_with_context_augmented: maps max |diff| / max = {rel:.3e}; keypoints identical: "
          f"{bool(np.array_equal(kp.numpy(), g9['keypoints']))}")
    np.testing.assert_allclose(maps9.numpy(), g9["maps"], rtol=1e-3, atol=1e-6)
    assert np.array_equal(kp.numpy(), g9["keypoints"])
    ptp_utils.collect_maps = real_collect
    print("INTEGRATION.md A: stages 2, 3 and 5 of main.py run on the build's load_ldm with the reference's own code")


if __name__ == "__main__":
    sys.exit(main())
