"""CPU tests of the host side: API surface, module-tree contract, C-ABI exports, layout rules."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from stablekeypoints_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "skp.h")).read()
    declared = sorted(set(re.findall(r"^int(?:64_t)?\s+(skp_\w+)\s*\(", hdr, flags=re.M)))
    assert len(declared) >= 9
    lib = N.lib()                                   # dlopen works without a GPU
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/skp.h but not exported"
        assert name in N.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.skp_abi_version() == N.ABI_VERSION
    # argument validation happens before any launch
    assert lib.skp_token_stats_f32(None, 0, 0, 1, 1.0, 1e-5, None, None, None, None) == -1


def test_library_reads_no_environment_and_tune_overrides_are_explicit():
    """Launch plans are functions of the shapes only: no getenv() in the library; the developer overrides go through
    skp_tune_set (tests / tools) and the package keeps to the documented switches (SKP_LIB_PATH, SKP_LAB_PATH, SKP_DIST_BACKEND,
    SKP_TUNABLEOP, SKP_ALLOW_SYNTHETIC)."""
    from stablekeypoints_amd import _native as N
    csrc = os.path.join(ROOT, "stablekeypoints_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f"{f} reads the environment"
    seen = set()
    for dp, _, files in os.walk(os.path.join(ROOT, "stablekeypoints_amd")):
        for f in files:
            if f.endswith(".py"):
                seen |= set(re.findall(r"[\"'](SKP_[A-Z0-9_]+)[\"']", open(os.path.join(dp, f)).read()))
    assert seen <= {"SKP_LIB_PATH", "SKP_LAB_PATH", "SKP_DIST_BACKEND", "SKP_TUNABLEOP", "SKP_ALLOW_SYNTHETIC"}, seen
    lib = N.lib()
    assert lib.skp_tune_get(b"wino_split") == 0 and lib.skp_tune_set(b"no_such_key", 1) == -2 and lib.skp_tune_set(b"wino_split", -1) == -1
    plan = lib.skp_conv3x3_f4_workspace(2, 640, 640, 32, 32)
    try:
        N.tune("wino_split", 3)
        assert lib.skp_tune_get(b"wino_split") == 3
        assert lib.skp_conv3x3_f4_workspace(2, 640, 640, 32, 32) == 3 * 2 * 640 * 32 * 32 * 4
    finally:
        N.tune("wino_split", 0)
    assert lib.skp_conv3x3_f4_workspace(2, 640, 640, 32, 32) == plan
    assert lib.skp_group_norm_onepass_ok(8, 320, 32, 64 * 64) == 1 and lib.skp_group_norm_onepass_ok(8, 128, 32, 512 * 512) == 0


def test_product_never_imports_oracle_and_has_no_cpu_fallback():
    pkg = os.path.join(ROOT, "stablekeypoints_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports oracle/"
    from stablekeypoints_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.token_stats(torch.rand(3, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.qk_logits(torch.rand(1, 16, 8), torch.rand(1, 3, 8), 2, 0.5)


def test_no_reference_source_in_repo():
    assert not os.path.exists(os.path.join(ROOT, "unsupervised_keypoints"))
    out = subprocess.run(["git", "-C", ROOT, "ls-files"], capture_output=True, text=True).stdout.split()
    assert not [f for f in out if f.endswith((".so", ".o"))], "built artefacts must stay out of history"


def test_sd15_module_tree_contract():
    """The reference's name-based patcher must find 18 CrossAttention modules in up_blocks (9 self + 9
    cross) and state-dict keys must match the 0.8.0 checkpoint layout."""
    from stablekeypoints_amd.ldm.unet import UNet2DConditionModel
    from stablekeypoints_amd import ptp_utils
    with torch.device("meta"):
        unet = UNet2DConditionModel()
    assert sum(p.numel() for p in unet.parameters()) == 859_520_964          # SD-1.x UNet
    keys = dict(unet.state_dict())
    assert tuple(keys["up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_k.weight"].shape) == (1280, 768)
    assert tuple(keys["up_blocks.2.attentions.0.transformer_blocks.0.attn2.to_q.weight"].shape) == (640, 640)
    assert tuple(keys["up_blocks.3.resnets.0.conv1.weight"].shape) == (320, 960, 3, 3)
    assert tuple(keys["down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight"].shape) == (2560, 320)
    assert "up_blocks.0.attentions.0.norm.weight" not in keys                   # UpBlock2D has no attention
    ctrl = ptp_utils.AttentionStore()
    ptp_utils.register_attention_control(unet, ctrl, feature_upsample_res=128)
    assert ctrl.num_att_layers == 18
    patched = [n for n, m in unet.named_modules() if m.__class__.__name__ == "CrossAttention" and "forward" in m.__dict__]
    assert len(patched) == 18 and all(n.startswith("up_blocks") for n in patched)
    heads = {n: m.heads for n, m in unet.named_modules() if m.__class__.__name__ == "CrossAttention"}
    assert set(heads.values()) == {8}
    with pytest.raises(AssertionError, match="No cross attention"):
        ptp_utils.register_attention_control(torch.nn.Sequential(torch.nn.Linear(2, 2)), ptp_utils.AttentionStore())


@pytest.fixture(scope="module")
def tiny_ldm():
    from stablekeypoints_amd.optimize_token import load_ldm
    return load_ldm("cpu", "tiny", feature_upsample_res=32)


def test_hook_gate_handles_and_early_exit(tiny_ldm):
    """First 4 cross layers with seq <= 32^2 are recorded (ptp_utils.py:508-512) as FusedAttn handles; the
    early exit leaves the same store; the map reduction itself refuses to run off-GPU."""
    from stablekeypoints_amd import ptp_utils
    from stablekeypoints_amd._maps import FusedAttn, collect_maps
    ldm, controllers, n = tiny_ldm
    assert n == 1 and list(controllers) == [torch.device("cpu")]
    ctrl = controllers[torch.device("cpu")]
    assert ctrl.num_att_layers == 18
    assert not any(p.requires_grad for p in ldm.unet.parameters())
    ctx = torch.randn(1, 9, 768, requires_grad=True)
    img = torch.rand(2, 3, 128, 128)
    for early in (False, True):
        ctrl.reset()
        noise, pred = ptp_utils.find_pred_noise(ldm, img, ctx, device="cpu", early_exit=early, controllers=controllers)
        recs = ctrl.step_store["attn"]
        assert len(recs) == 4 and all(isinstance(r, FusedAttn) for r in recs)
        assert [r.q.shape[1] for r in recs] == [16, 16, 16, 64]             # 3x(4x4) then 8x8 at a 16^2 latent
        assert [r.shape for r in recs] == [(2 * 4, 32 * 32, 9)] * 4
        assert all(r.k.requires_grad for r in recs)
        assert (pred is None) == early and ctrl.stop_after is None
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        collect_maps(ctrl, upsample_res=-1)
    ctrl.reset()
    assert ctrl.step_store == {"attn": []} and ctrl.cur_att_layer == 0


def test_attention_store_api_surface():
    from stablekeypoints_amd.ptp_utils import AttentionControl, AttentionStore
    s = AttentionStore()
    assert isinstance(s, AttentionControl) and AttentionStore.get_empty_store() == {"attn": []}
    assert s.num_att_layers == -1 and s.cur_step == 0 and s.num_uncond_att_layers == 0
    x = object()
    assert s({"attn": x}, True, "up") is x and s.step_store["attn"] == [x]
    assert s.step_callback(3) == 3 and s.between_steps() is None
    s.reset()
    assert s.step_store["attn"] == []
    with pytest.raises(TypeError):
        AttentionControl()


def test_affine_transform_matches_reference_golden(golden):
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from oracle.fixtures import seeded
    g = golden("g7_gauss_affine.npz")
    tr = RandomAffineWithInverse(degrees=15, scale=(0.8, 1.0), translate=(0.25, 0.25))
    thetas = torch.cat([tr.create_affine_matrix(11.0, 0.87, (0.13, -0.21)), tr.create_affine_matrix(-14.0, 0.95, (-0.2, 0.05))])
    img = seeded((2, 3, 20, 28), 77).abs()
    w = tr(img, theta=thetas)
    torch.testing.assert_close(w, torch.from_numpy(g["warp"]), rtol=2e-5, atol=2e-7)
    torch.testing.assert_close(tr.inverse(w), torch.from_numpy(g["unwarp"]), rtol=2e-5, atol=2e-7)
    torch.manual_seed(1234)
    tr(img)
    torch.testing.assert_close(tr.last_params["theta"], torch.from_numpy(g["theta_seed1234"]), rtol=0, atol=0)
    from stablekeypoints_amd.ops import invert_affine
    inv = torch.tensor(invert_affine(thetas[0].reshape(-1).tolist())).reshape(2, 3)
    torch.testing.assert_close(inv, tr.invert(thetas[:1])[0], rtol=1e-5, atol=1e-6)


def test_gaussian_and_scheduler_match_oracle(golden):
    from stablekeypoints_amd.optimize_token import gaussian_circle, gaussian_circles
    from stablekeypoints_amd.ldm.scheduler import DDIMScheduler
    from oracle import ref_path as R
    g = golden("g7_gauss_affine.npz")
    pos = torch.tensor([[[0.3, 0.7], [0.5, 0.5], [0.02, 0.98]], [[0.9, 0.1], [0.25, 0.75], [0.6, 0.4]]])
    torch.testing.assert_close(gaussian_circle(pos[0], 24, 2.0, "cpu"), torch.from_numpy(g["gaussian_circle"]))
    torch.testing.assert_close(gaussian_circles(pos, 24, 3.0, "cpu"), torch.from_numpy(g["gaussian_circles"]))
    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    s.set_timesteps(50)
    assert torch.equal(s.timesteps, R.ddim_timesteps())
    x, n = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    torch.testing.assert_close(s.add_noise(x, n, s.timesteps[-1]), R.add_noise(x, n, 0))


def test_optimize_embedding_rejects_bad_batch():
    from stablekeypoints_amd.optimize import optimize_embedding, default_args
    from stablekeypoints_amd.ptp_utils import AttentionStore
    with pytest.raises(ValueError, match="multiple of the data-parallel width"):
        optimize_embedding(None, default_args(batch_size=0), {torch.device("cpu"): AttentionStore()}, 1)
    a = default_args()
    assert a.num_tokens == 500 and a.feature_upsample_res == 128 and a.top_k == 10 and a.lr == 5e-3


def test_c_abi_rejects_bad_arguments_before_launching():
    """Every entry point validates pointers/sizes on the host and returns SKP_E_* (< 0) without touching the GPU."""
    import ctypes as C
    from stablekeypoints_amd import _native as N
    lib = N.lib()
    null = None
    assert lib.skp_qk_logits_f32(null, null, null, 1, 1, 8, 77, 256, 160, 0.1, null) == -1
    assert lib.skp_gemm_nt_f32(null, null, null, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, null) == -1
    one = C.c_float(0.0)
    p = C.cast(C.pointer(one), C.c_void_p)
    sp, _k = N.ptr_array([p.value]); si, _k2 = N.int_array([16])
    assert lib.skp_attn_map_fwd_f32(sp, si, 1, 1, 8, 500, 128, p, p, null) == -2          # T > 128: use the _ex form
    assert lib.skp_attn_map_fwd_f32(sp, si, 9, 1, 8, 77, 128, p, p, null) == -2           # too many layers
    assert lib.skp_attn_map_fwd_ex_f32(sp, si, 1, 1, 8, 77, 128, p, p, null, 64, 0, 0, null) == -1   # ldt < NT
    assert lib.skp_attn_map_bwd_workspace(si, 0, 1, 8, 77, 128) < 0
    assert lib.skp_attn_map_bwd_workspace(si, 1, 2, 8, 77, 128) == 2 * 8 * 128 * 16 * 80 * 4
    assert lib.skp_cross_attn_fwd_f32(p, p, p, p, p, 1, 1, 8, 64, 77, 48, 0.1, null) == -2    # unsupported head dim
    assert lib.skp_cross_attn_fwd_f32(p, p, p, p, p, 2, 3, 8, 64, 77, 40, 0.1, null) == -1    # Bk not in {1,B}
    assert lib.skp_self_attn_fwd_f32(p, p, p, p, p, 1, 8, 64, 24, 0.1, null) == -2
    assert lib.skp_group_norm_fwd_f32(p, null, p, p, p, p, p, p, 1, 30, 32, 16, 1e-5, 1, null) == -2   # C % G
    assert lib.skp_select_tokens(p, p, 77, 128, 25, 1, p, p, null) == -2                  # top_k < 2
    assert lib.skp_token_stats_f32(p, 77, 128, 9, 2.0, 1e-5, p, null, null, null) == -2         # too many subjects
    assert lib.skp_losses_fwd_f32(p, p, p, 10, 77, 128, p, 1, 2.0, None, p, p, p, p, null) == -1
    # round 3 entry points
    assert lib.skp_add_layer_norm_ok(320) == 1 and lib.skp_add_layer_norm_ok(1280) == 1 and lib.skp_add_layer_norm_ok(30) == 0
    assert lib.skp_add_layer_norm_fwd_f32(null, null, p, p, null, p, p, 4, 320, 1e-5, null) == -1      # no input
    assert lib.skp_add_layer_norm_fwd_f32(p, p, p, p, null, p, p, 4, 320, 1e-5, null) == -1         # d without a place for d + h
    assert lib.skp_add_layer_norm_fwd_f32(null, p, p, p, null, p, p, 4, 30, 1e-5, null) == -2        # row width without a plan
    assert lib.skp_add_layer_norm_bwd_f32(p, null, p, p, p, null, 4, 320, null) == -1
    assert lib.skp_unwarp_accumulate_f32(null, p, 1, 1, 32, 64, p, p, 1, null) < 0
    assert lib.skp_conv3x3_f4_gn_ok(8, 128, 128, 512, 512) == 1 and lib.skp_conv3x3_f4_gn_ok(8, 512, 512, 64, 64) == 1
    assert lib.skp_conv3x3_f4_gn_ok(8, 640, 640, 32, 32) == 0 and lib.skp_conv3x3_f4_gn_ok(8, 1280, 640, 16, 16) == 0   # <= 4 channel groups, unsplit launches
    assert lib.skp_conv3x3_f4_gn_f32(p, p, null, null, p, null, null, 8, 128, 128, 512, 512, null) == -1              # no coefficients


def test_tuning_module_is_inert_without_a_gpu():
    """The GEMM algorithm file is only read on a GPU box; on CPU `enable()` is a no-op and never raises."""
    from stablekeypoints_amd import tuning
    import os
    assert os.path.exists(os.path.join(os.path.dirname(tuning.__file__), "tunableop_gfx950.csv"))
    if not torch.cuda.is_available():
        assert tuning.enable() is False


def test_conv3x3_shape_rules():
    """Which 3x3 convolutions go to the Winograd kernels (pure host logic, no launches)."""
    from stablekeypoints_amd import ops
    assert ops.conv3x3_supported((8, 128, 512, 512), (128, 128, 3, 3))
    assert not ops.conv3x3_supported((8, 4, 64, 64), (320, 4, 3, 3))            # conv_in: 4 channels -> library
    assert not ops.conv3x3_supported((8, 128, 64, 64), (128, 128, 1, 1))
    assert ops.conv3x3_f4_ok((8, 320, 64, 64), (320, 320, 3, 3)) == (ops.CONV3X3_MODE == "f4")
    assert not ops.conv3x3_f4_ok((8, 320, 66, 64), (320, 320, 3, 3))            # H % 4 != 0 -> F(2x2,3x3)
    assert ops.conv3x3_wanted((8, 320, 66, 64), (320, 320, 3, 3)) == (ops.CONV3X3_MODE != "lib")
    assert not ops.conv3x3_wanted((1, 32, 4, 4), (32, 32, 3, 3))               # too few tiles -> library


def test_checkpoint_directory_loading_is_strict(tmp_path):
    """`load_ldm(<dir>)`: both accepted layouts load key by key; a missing / mismatched / unexpected key raises (only the VAE
    decoder half is tolerated); a hub id that is not a local directory raises instead of silently building random weights."""
    import json
    from safetensors.torch import save_file
    from stablekeypoints_amd.ldm.pipeline import ARCHS, StableDiffusionPipeline
    from stablekeypoints_amd.optimize_token import load_ldm
    unet, vae = StableDiffusionPipeline.build("tiny-sdxl", seed=7)
    # (a) diffusers layout: <dir>/unet/{config.json, diffusion_pytorch_model.safetensors}, <dir>/vae/... (+ decoder keys)
    d = tmp_path / "some-xl-checkpoint"
    (d / "unet").mkdir(parents=True); (d / "vae").mkdir()
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in ARCHS["tiny-sdxl"]["unet"].items()}
    (d / "unet" / "config.json").write_text(json.dumps(cfg))
    save_file({k: v.contiguous() for k, v in unet.state_dict().items()}, str(d / "unet" / "diffusion_pytorch_model.safetensors"))
    vsd = {k: v.contiguous() for k, v in vae.state_dict().items()}
    # the real VAE tree of this topology is wider than the sd15-shaped default: store it as .bin with decoder leftovers
    vsd["decoder.conv_in.weight"] = torch.zeros(4, 4, 3, 3)
    vsd["post_quant_conv.weight"] = torch.zeros(4, 4, 1, 1)
    torch.save(vsd, str(d / "vae" / "diffusion_pytorch_model.bin"))
    # the VAE of `tiny-sdxl` is reduced-width; the loader builds the VAE of the guessed arch ("xl" in the name -> sdxl, full width)
    with pytest.raises(RuntimeError, match="shape mismatches"):
        load_ldm("cpu", str(d), feature_upsample_res=32)
    # (b) flat layout with matching trees: unet.pt / vae.pt of the tiny model under a tiny-named directory
    d2 = tmp_path / "tiny-sdxl"
    d2.mkdir()
    torch.save(unet.state_dict(), str(d2 / "unet.pt"))
    torch.save(vae.state_dict(), str(d2 / "vae.pt"))
    import stablekeypoints_amd.ldm.pipeline as P
    orig = P.guess_arch
    P.guess_arch = lambda name: "tiny-sdxl"
    try:
        ldm, _, _ = load_ldm("cpu", str(d), feature_upsample_res=32)          # diffusers layout, decoder keys tolerated
        assert not ldm.synthetic_weights and ldm.unet.config["cross_attention_dim"] == 128
        assert torch.equal(ldm.vae.encoder.conv_in.weight, vae.encoder.conv_in.weight)
        ldm, _, _ = load_ldm("cpu", str(d2), feature_upsample_res=32)
        assert not ldm.synthetic_weights
        for (k, a_), (_, b_) in zip(ldm.unet.state_dict().items(), unet.state_dict().items()):
            assert torch.equal(a_, b_), k
        sd = unet.state_dict(); sd.pop("conv_in.bias")
        torch.save(sd, str(d2 / "unet.pt"))
        with pytest.raises(RuntimeError, match="1 missing"):
            load_ldm("cpu", str(d2), feature_upsample_res=32)
        sd = unet.state_dict(); sd["surprise.weight"] = torch.zeros(1)
        torch.save(sd, str(d2 / "unet.pt"))
        with pytest.raises(RuntimeError, match="1 unexpected"):
            load_ldm("cpu", str(d2), feature_upsample_res=32)
    finally:
        P.guess_arch = orig
    with pytest.raises(FileNotFoundError, match="not a local checkpoint directory"):
        load_ldm("cpu", "sd-legacy/stable-diffusion-v1-5", feature_upsample_res=32)


def test_sd2x_sdxl_module_tree_contract():
    """Published parameter counts of the SD-2.1 and SDXL-base UNets, hooked-layer census, embedding width."""
    from stablekeypoints_amd.ldm.pipeline import ARCHS
    from stablekeypoints_amd.ldm.unet import UNet2DConditionModel
    from stablekeypoints_amd import ptp_utils
    with torch.device("meta"):
        sd21 = UNet2DConditionModel(**ARCHS["sd21"]["unet"])
        sdxl = UNet2DConditionModel(**ARCHS["sdxl"]["unet"])
    assert sum(p.numel() for p in sd21.parameters()) == 865_910_724
    assert sum(p.numel() for p in sdxl.parameters()) == 2_567_463_684
    k21, kxl = dict(sd21.state_dict()), dict(sdxl.state_dict())
    assert tuple(k21["up_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape) == (1280, 1024)
    assert tuple(k21["up_blocks.1.attentions.0.proj_in.weight"].shape) == (1280, 1280)          # linear projection
    assert tuple(kxl["up_blocks.0.attentions.0.transformer_blocks.9.attn2.to_k.weight"].shape) == (1280, 2048)
    assert tuple(kxl["add_embedding.linear_1.weight"].shape) == (1280, 2816)
    assert "down_blocks.0.attentions.0.norm.weight" not in kxl                                    # SDXL starts with a DownBlock2D
    for net, n_up in ((sd21, 18), (sdxl, 2 * (3 * 10 + 3 * 2))):
        c = ptp_utils.AttentionStore()
        ptp_utils.register_attention_control(net, c, feature_upsample_res=128)
        assert c.num_att_layers == n_up
    heads = {m.heads for n, m in sdxl.named_modules() if m.__class__.__name__ == "CrossAttention"}
    assert heads == {10, 20}
    assert tuple(ptp_utils.init_random_noise("cpu", 77, sdxl.config["cross_attention_dim"]).shape) == (1, 77, 2048)


def test_linear_and_qkv_dispatch_rules_on_cpu():
    """Host logic of the projection helpers: on CPU tensors they are the library ops."""
    import torch
    from stablekeypoints_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 10, 64, generator=g, requires_grad=True)
    ws = [torch.randn(64, 64, generator=g) for _ in range(3)]
    b = torch.randn(64, generator=g)
    q, k, v = ops.qkv_proj(x, *ws)
    for a, w in zip((q, k, v), ws):
        assert torch.equal(a, torch.nn.functional.linear(x, w))
    y = ops.conv1x1_nobias(torch.randn(2, 8, 3, 5, generator=g), torch.randn(4, 8, 1, 1, generator=g))
    assert y.shape == (2, 4, 3, 5)


def test_constant_time_path_is_kept_and_invalidated():
    """`fused._time_path`: with a host timestep and a frozen time MLP the UNet hands every step the SAME time-embedding tensor
    (the resnets key their kept norm2 offsets on it); another timestep, an in-place weight change or a parameter that takes
    gradients give a fresh one; values equal the module's own `time_path` bit for bit."""
    from stablekeypoints_amd.ldm.fused import _norm2_offset, fuse_norms
    from stablekeypoints_amd.ldm.unet import ResnetBlock2D
    from stablekeypoints_amd.optimize_token import load_ldm
    ldm, _, _ = load_ldm("cpu", "tiny", feature_upsample_res=32)
    unet = ldm.unet
    fuse_norms(unet)                                              # (load_ldm only does this on a GPU)
    own = type(unet).time_path
    x = torch.randn(2, 4, 16, 16)
    t = ldm.scheduler.timesteps[-1]
    a, b = unet.time_path(x, t.repeat(2)), unet.time_path(x, t.repeat(2))
    assert a is b and not a.requires_grad and torch.equal(a, own(unet, x, t.repeat(2)))
    c = unet.time_path(x, ldm.scheduler.timesteps[0].repeat(2))
    assert c is not a and not torch.equal(c, a)
    assert unet.time_path(x[:1], t.repeat(1)).shape[0] == 1      # rows are part of the key
    res = next(m for m in unet.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None)
    o1, o2 = _norm2_offset(res, 2, a), _norm2_offset(res, 2, a)
    assert o1 is o2 and o1.is_contiguous()
    assert torch.equal(o1, res.conv1.bias[None] + res.time_emb_proj(torch.nn.functional.silu(a)))
    assert _norm2_offset(res, 2, c) is not o1                    # another embedding tensor: recomputed
    with torch.no_grad():
        unet.time_embedding.linear_1.weight.mul_(1.5)            # version bump
        res.conv1.bias.add_(1.0)
    d = unet.time_path(x, t.repeat(2))
    assert d is not a and torch.equal(d, own(unet, x, t.repeat(2))) and not torch.equal(d, a)
    assert not torch.equal(_norm2_offset(res, 2, a), o1)         # the block's own parameters changed
    unet.time_embedding.linear_1.weight.requires_grad_(True)
    e1, e2 = unet.time_path(x, t.repeat(2)), unet.time_path(x, t.repeat(2))
    assert e1 is not e2 and e1.requires_grad


# ---------------------------------------------------------------------------------------------------------------------
# round 4
# ---------------------------------------------------------------------------------------------------------------------
def test_lab_library_is_separate_and_exports_its_header():
    """Measurement aids / experiments are NOT in the product library: tools/csrc/libskp_lab.so exports what
    tools/csrc/skp_lab.h declares, libskp_hip.so exports none of it."""
    from stablekeypoints_amd import _native as N
    lab_dir = os.path.join(ROOT, "tools", "csrc")
    if not os.path.exists(N.LAB_PATH):
        subprocess.run(["make", "-C", lab_dir, "-j2"], check=True)
    hdr = open(os.path.join(lab_dir, "skp_lab.h")).read()
    declared = sorted(set(re.findall(r"^int\s+(skp_\w+)\s*\(", hdr, flags=re.M)))
    assert declared == sorted(N.LAB_SIGNATURES) and len(declared) >= 1
    lab, lib = N.lab(), N.lib()
    for name in declared:
        assert hasattr(lab, name) and not hasattr(lib, name), name


def test_fused_attn_handle_is_a_tensor_duck_type():
    """The reference's own `optimize.collect_maps` op sequence (optimize.py:52-75) on `FusedAttn` handles: reshape, index,
    permute, interpolate, stack + mean -- equal to the same ops on the materialised (B*h, R^2, T) tensor; the handle
    materialises once; shape / len / device answer without materialising."""
    import torch.nn.functional as F
    from stablekeypoints_amd._maps import FusedAttn, collect_maps
    from stablekeypoints_amd.ptp_utils import AttentionStore
    g = torch.Generator().manual_seed(3)
    recs = [FusedAttn(torch.randn(2, s * s, 32, generator=g), torch.randn(1, 7, 32, generator=g), 4, 0.35, 12) for s in (4, 4, 6)]
    assert recs[0].shape == (8, 144, 7) and len(recs[0]) == 8 and recs[0]._mat is None and recs[0].device.type == "cpu"
    idx = torch.tensor([3, 0, 3])

    def reference_order(store):
        out = []
        for data in store:
            data = data.reshape(data.shape[0], int(data.shape[1] ** 0.5), int(data.shape[1] ** 0.5), data.shape[2])
            data = data[:, :, :, idx].permute(0, 3, 1, 2)
            out.append(F.interpolate(data, size=(20, 20), mode="bilinear", align_corners=False))
        return torch.stack(out, dim=0).mean(dim=(0, 1))

    got = reference_order(recs)
    want = reference_order([r.materialize() for r in recs])
    assert torch.equal(got, want) and got.shape == (3, 20, 20)
    first = recs[0]._mat
    assert first is not None and recs[0].materialize() is first                 # cached
    torch.testing.assert_close(first.sum(-1), torch.ones(8, 144))               # softmax over the tokens
    assert torch.equal(torch.stack(recs[:2]), torch.stack([recs[0]._mat, recs[1]._mat]))
    # the package's own collect_maps on a store of handles that live on the HOST takes the materialised route
    ctrl = AttentionStore()
    for r in recs[:2]:
        ctrl.step_store["attn"].append(r.materialize())
    m = collect_maps(ctrl, upsample_res=-1, layers=[0, 1])
    assert m.shape == (7, 12, 12) and len(ctrl.step_store["attn"]) == 0


def test_wide_map_gate_mirrors_the_kernel_limits():
    from stablekeypoints_amd import _native as N, ops
    lib = N.lib()

    def ok(sides, T, R):
        si, _k = N.int_array(sides)
        return lib.skp_attn_map_fwd_wide_ok(si, len(sides), T, R)
    assert ok([16, 16, 16, 32], 500, 128) == 1 and ok([64], 1000, 512) == 1
    assert ok([65], 500, 128) == 0                      # layer side beyond the kernel's 64
    assert ok([16], 500, 100) == 0                      # R % 32
    assert ok([16], 1025, 128) == 0 and ok([16], 500, 2048) == 0       # token count; R * (R / 32) tiles > 65535
    assert ops.map_wide_supported(500, 128, [16, 32]) and not ops.map_wide_supported(500, 128, [96])
    assert not ops.map_wide_supported(77, 128, [16])    # T <= 128 is the narrow kernel's


def _kp_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from stablekeypoints_amd import dist as D, keypoint_regressor as KR
    if world > 1:
        D.init_from_env("gloo")
    src, tgt, vis = _run_fake_keypoints(KR)
    torch.save({"src": src, "tgt": tgt}, os.path.join(out, f"w{world}r{rank}.pt"))
    if world > 1:
        D.barrier()
        dist.destroy_process_group()


def _run_fake_keypoints(KR):
    """precompute_all_keypoints with the network replaced by a deterministic host function of (image, thetas, noise)."""
    from types import SimpleNamespace
    n_img, K, n_aug = 7, 3, 2

    def fake_aug(ldm, images, context, indices, controllers=None, thetas=None, noise=None, upscale_size=512, **kw):
        m = images.shape[0]
        maps = torch.zeros(m, K, 16, 16)
        for i in range(m):
            for k in range(K):
                v = float(images[i].sum() + thetas[i * n_aug:(i + 1) * n_aug].sum() + noise[i * n_aug:(i + 1) * n_aug].sum()) + k
                maps[i, k, int(abs(v) * 7) % 16, int(abs(v) * 13) % 16] = 1.0
        return maps

    KR.run_images_with_context_augmented = fake_aug
    KR.keypoints_from_maps = lambda mp, strategy="argmax": torch.stack(
        [mp.flatten(1).argmax(1) // 16, mp.flatten(1).argmax(1) % 16], dim=-1).float() / 16
    g = torch.Generator().manual_seed(5)
    data = [{"img": torch.rand(3, 8, 8, generator=g), "kpts": torch.rand(4, 2, generator=g)} for _ in range(n_img)]
    args = SimpleNamespace(max_num_points=6, augmentation_iterations=n_aug, layers=[0], noise_level=-1, augment_degrees=1,
                           augment_scale=[1, 1], augment_translate=[0, 0], max_loc_strategy="argmax", images_per_forward=2)
    draws = (torch.randperm(n_img, generator=g).tolist(), torch.rand(n_img * n_aug, 4, 1, 1, generator=g),
             torch.rand(n_img * n_aug, 2, 3, generator=g))
    return KR.precompute_all_keypoints(None, torch.zeros(1, 4, 8), torch.arange(K), args, {torch.device("cpu"): None}, 1,
                                       dataset=data, draws=draws)


def test_precompute_all_keypoints_rank_sharding_gloo(tmp_path):
    """Dataset-level keypoint driver, host logic: positions of the shuffled order are dealt `p % world`, locations are
    all-gathered back into loader order (uneven split: 6 images over 2 ranks x groups of 2 ... and `max_num_points` cuts
    the pass short), targets follow the same order -- two gloo ranks == one process."""
    import torch.multiprocessing as mp
    _kp_worker(0, 1, 0, str(tmp_path))
    one = torch.load(tmp_path / "w1r0.pt")
    assert one["src"].shape == (6, 3, 2) and one["tgt"].shape == (6, 4, 2)
    mp.spawn(_kp_worker, args=(2, 29300 + os.getpid() % 200, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "w2r0.pt"), torch.load(tmp_path / "w2r1.pt")
    for key in ("src", "tgt"):
        assert torch.equal(a[key], b[key]) and torch.equal(a[key], one[key]), key


def test_bench_self_launches_when_no_launcher_environment():
    """`python bench.py --gpus N` (N > 1) without torchrun's environment must become the launcher itself instead of dying on
    WORLD_SIZE != N: the command is the driver's own N > 1 line, and a rank that dies (here: no GPU in this container) makes
    the whole call exit non-zero with no JSON line on stdout."""
    import subprocess
    import bench
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "5"], port=29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "8", "--steps", "5"]
    p1 = int(bench.self_launch_command(2, [])[bench.self_launch_command(2, []).index("--master-port") + 1])
    assert 1024 < p1 < 65536
    if torch.cuda.is_available():
        return                                                     # the GPU half is tests/test_round5_gpu.py
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--steps", "1",
                          "--warmup", "0", "--cpu-baseline", "off"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "torch.distributed.run" in out.stderr and "no GPU visible" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_fused_attn_materialize_under_autograd_names_the_missing_rebind():
    """A handle whose q / k carry gradient history must not be materialised while autograd records (the reference's
    `optimize_embedding` kept on a re-bound `load_ldm`): the error names the second re-bind instead of autograd's generic
    'does not require grad' at `loss.backward()`; no_grad callers (inference) and detached handles are unaffected; unknown
    attributes do not materialise."""
    from stablekeypoints_amd._maps import FusedAttn
    g = torch.Generator().manual_seed(1)
    q = torch.randn(1, 16, 8, generator=g).requires_grad_(True)
    k = torch.randn(1, 5, 8, generator=g)
    rec = FusedAttn(q * 2.0, k, 2, 0.5, 8)
    with pytest.raises(RuntimeError, match="re-bind .*optimize_embedding"):
        rec.reshape(2, 8, 8, 5)
    assert rec._mat is None
    with torch.no_grad():
        assert rec.reshape(2, 8, 8, 5).shape == (2, 8, 8, 5)
    rec2 = FusedAttn(q.detach(), k, 2, 0.5, 8)
    assert rec2.ndim == 3 and rec2.numel() == 2 * 64 * 5 and rec2.requires_grad is False and rec2._mat is None
    assert not hasattr(rec2, "no_such_tensor_method") and rec2._mat is None       # a probe does not build the tensor
    with pytest.raises(AttributeError, match="forwarded"):
        rec2.bogus
    import collections
    Pair = collections.namedtuple("Pair", "a b")
    out = torch.stack(Pair(rec2, rec2))                                           # namedtuple argument: rebuilt as a tuple
    assert out.shape == (2, 2, 64, 5)


def test_group_indices_order_is_loader_independent(tmp_path):
    """The index stream of `optimize_embedding` (epoch-wise shuffle, rank sharding, injected order) and `GroupLoader`'s
    synchronous mode: ranks partition every epoch's permutation, groups follow the accumulation count, images are the
    dataset's items in that order."""
    from stablekeypoints_amd.optimize import GroupLoader, SyntheticImages, _group_indices, default_args
    a = default_args(num_steps=5, seed=3)
    r0 = list(_group_indices(a, 10, None, 2, 2, 0, 2))
    r1 = list(_group_indices(a, 10, None, 2, 2, 1, 2))
    assert [it for it, _ in r0] == [0, 2, 4, 6, 8] and all(len(i) == 2 for _, i in r0)
    first_epoch = [i for _, idx in r0[:2] for i in idx] + r0[2][1][:1] + [i for _, idx in r1[:2] for i in idx] + r1[2][1][:1]
    assert sorted(first_epoch) == list(range(10))                   # 5 per rank per epoch: a partition of the permutation
    assert list(_group_indices(a, 10, None, 2, 2, 0, 2)) == r0      # deterministic in args.seed
    one = list(_group_indices(a, 10, None, 2, 1, 0, 2))             # images_per_forward = 1: same images, one per group
    assert [i for _, idx in one for i in idx] == [i for _, idx in r0 for i in idx]
    inj = list(_group_indices(a, 10, ([7, 7, 1, 2, 3, 4, 5, 6, 8, 9],), 2, 2, 0, 1))
    assert inj[0] == (0, [7, 7]) and inj[4] == (8, [8, 9])
    ds = SyntheticImages(n=10, size=8)
    got = list(GroupLoader(ds, _group_indices(a, 10, None, 2, 2, 0, 2), "cpu", workers=4))
    assert [(it, idx) for it, idx, _ in got] == r0
    assert all(torch.equal(im, torch.stack([ds[i]["img"] for i in idx])) for _, idx, im in got)


def test_weight_stack_is_cached_per_parameter_list_and_follows_versions():
    """Stacked copies of frozen projection weights (q | k | v of a self-attention block, k / v of the cross-attention layers of one
    width): made once, re-made when a parameter is written, dropped when its model is gone."""
    import gc
    import torch
    from stablekeypoints_amd import ops
    a, b = torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(4, 3))
    s1 = ops.weight_stack([a, b])
    assert s1.shape == (2, 4, 3) and torch.equal(s1[1], b.detach())
    assert ops.weight_stack([a, b]) is s1
    with torch.no_grad():
        b.mul_(2.0)                                               # version bump: the stack must follow
    s2 = ops.weight_stack([a, b])
    assert s2 is not s1 and torch.equal(s2[1], b.detach())
    n = len(ops._QKV_CACHE)
    b.data = b.data.clone()                                       # storage moved (`.to()`, dtype change): the entry is REPLACED, not added
    s3 = ops.weight_stack([a, b])
    assert s3 is not s2 and len(ops._QKV_CACHE) == n
    del a, b, s1, s2, s3
    gc.collect()
    c = torch.nn.Parameter(torch.randn(2, 2))
    ops.weight_stack([c, c])                                      # inserting prunes entries whose owner is gone
    assert len(ops._QKV_CACHE) <= n
