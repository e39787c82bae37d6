"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and vs the golden vectors that were
captured from the reference.  Tolerances: probabilities/maps rtol 1e-3 (BASELINE.json north_star)
with atol 1e-6 for near-zero probabilities; integer outputs (arg-max, selected tokens) bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_path as R
from oracle.fixtures import (HOOK_CASES, STACK_CASE, SEL_CASE, E2E_CASE, attention_weights, seeded,
                             selection_maps)

pytestmark = pytest.mark.gpu

MAP_TOL = dict(rtol=1e-3, atol=1e-6)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from stablekeypoints_amd import ops as o
    o.N.lib()                     # raises if libskp_hip.so is missing: no fallback
    return o


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_gemm_nt_mfma(ops):
    g = torch.Generator().manual_seed(0)
    for (M, N, K, Z0, Z1) in [(77, 256, 160, 2, 8), (33, 31, 10, 1, 3), (128, 64, 7, 1, 1), (5, 200, 80, 3, 2)]:
        A = torch.randn(Z0, Z1, M, K, generator=g)
        B = torch.randn(Z0, Z1, N, K, generator=g)
        ref = torch.einsum("abmk,abnk->abmn", A.double(), B.double()) * 0.37
        Ad, Bd = A.cuda(), B.cuda()
        C = torch.empty(Z0, Z1, M, N, device="cuda")
        ops._gemm_nt(Ad, Bd, C, M, N, K, Z0, Z1, (Z1 * M * K, M * K, K, 1), (Z1 * N * K, N * K, K, 1),
                     (Z1 * M * N, M * N, N), 0.37)
        torch.testing.assert_close(C.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    # transposed-operand form (k strided), as used by the backward products
    M, N, K = 40, 24, 50
    A = torch.randn(K, M, generator=g); B = torch.randn(K, N, generator=g)
    C = torch.empty(M, N, device="cuda")
    ops._gemm_nt(A.cuda(), B.cuda(), C, M, N, K, 1, 1, (0, 0, 1, M), (0, 0, 1, N), (0, 0, N), 1.0)
    torch.testing.assert_close(C.cpu().double(), A.double().t() @ B.double(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", list(HOOK_CASES))
def test_g1_hook_probs_vs_reference_golden(ops, golden, name):
    g = golden("g1_hook.npz")
    c = HOOK_CASES[name]
    wq, wk, wv, wo, bo = attention_weights(c["C"], c["ctx_dim"], c["seed"])
    x = seeded((c["B"], c["s"] ** 2, c["C"]), c["seed"] + 100)
    ctx = seeded((c["B"], c["T"], c["ctx_dim"]), c["seed"] + 200)
    q = (x @ wq.t()).cuda()
    k = (ctx @ wk.t()).cuda()
    scale = (c["C"] // c["heads"]) ** -0.5
    p = ops.materialize_probs(q, k, c["heads"], scale, c["R"]).cpu()
    assert p.shape == (c["B"] * c["heads"], c["R"] ** 2, c["T"])
    if c["full"]:
        torch.testing.assert_close(p, t(g[name + "/p_up"]), **MAP_TOL)
    else:
        torch.testing.assert_close(p.reshape(-1)[:: c["stride"]], t(g[name + "/p_up_strided"]), **MAP_TOL)
    torch.testing.assert_close(p.sum(-1), torch.ones(p.shape[:2]), rtol=1e-5, atol=1e-5)


def _stack_qk(case, ctx, view=0):
    qs, ks, scales = [], [], []
    for i, (sl, Cl) in enumerate(case["layers"]):
        wq, wk, *_ = attention_weights(Cl, case["ctx_dim"], case["seed"] + i)
        x = seeded((1, sl * sl, Cl), case["seed"] + 100 + 10 * view + i)
        qs.append((x @ wq.t()).cuda())
        ks.append(ctx @ wk.t().to(ctx.device))
        scales.append((Cl // case["heads"]) ** -0.5)
    return qs, ks, scales


def test_g2_fused_map_vs_reference_collect_maps(ops, golden):
    g = golden("g2_collect_maps.npz")
    c = STACK_CASE
    ctx = seeded((1, c["T"], c["ctx_dim"]), c["seed"] + 200).cuda()
    qs, ks, scales = _stack_qk(c, ctx)
    M = ops.attn_map(qs, ks, c["heads"], scales, c["R"])[0].cpu()
    torch.testing.assert_close(M, t(g["res-1"]), **MAP_TOL)
    M02 = ops.attn_map([qs[0], qs[2]], [ks[0], ks[2]], c["heads"], [scales[0], scales[2]], c["R"])[0].cpu()
    torch.testing.assert_close(M02, t(g["res-1_layers02"]), **MAP_TOL)
    torch.testing.assert_close(M.sum(0), torch.ones(c["R"], c["R"]), rtol=1e-5, atol=1e-5)


def test_attn_map_backward_vs_oracle_autograd(ops):
    """d(sum(M*W))/d(q,k) of the fused op vs fp64 autograd through the reference formulation."""
    c = STACK_CASE
    heads, Rr, T = c["heads"], c["R"], c["T"]
    B = 2
    g = torch.Generator().manual_seed(5)
    qs = [torch.randn(B, sl * sl, Cl, generator=g) for sl, Cl in c["layers"]]
    ks = [torch.randn(1, T, Cl, generator=g) for sl, Cl in c["layers"]]
    W = torch.randn(B, T, Rr, Rr, generator=g)
    scales = [(Cl // heads) ** -0.5 for _, Cl in c["layers"]]
    # oracle (fp64): upsample q (not logits), per-head softmax, mean -- the reference's order of ops
    qd = [q.double().requires_grad_(True) for q in qs]
    kd = [k.double().requires_grad_(True) for k in ks]
    maps = []
    for q, k, (sl, Cl), sc in zip(qd, kd, c["layers"], scales):
        qi = q.reshape(B, sl, sl, Cl).permute(0, 3, 1, 2)
        qu = torch.nn.functional.interpolate(qi, size=(Rr, Rr), mode="bicubic", align_corners=False)
        qu = R.split_heads(qu.permute(0, 2, 3, 1).reshape(B, Rr * Rr, Cl), heads)
        kk = R.split_heads(k.expand(B, -1, -1), heads)
        p = (torch.einsum("bid,bjd->bij", qu, kk) * sc).softmax(-1)          # (B*h, R2, T)
        maps.append(p.reshape(B, heads, Rr, Rr, T).permute(0, 1, 4, 2, 3))
    Mref = torch.stack(maps, 0).mean(dim=(0, 2))                              # [B,T,R,R]
    (Mref * W.double()).sum().backward()
    qg = [q.cuda().requires_grad_(True) for q in qs]
    kg = [k.cuda().requires_grad_(True) for k in ks]
    M = ops.attn_map(qg, kg, heads, scales, Rr)
    torch.testing.assert_close(M.detach().cpu().double(), Mref.detach(), rtol=1e-3, atol=1e-6)
    (M * W.cuda()).sum().backward()
    for a, b in zip(qg + kg, qd + kd):
        ref = b.grad
        torch.testing.assert_close(a.grad.cpu().double(), ref, rtol=2e-3, atol=2e-5 * ref.abs().max().item())


def test_g3_token_stats_and_selection(ops, golden):
    g = golden("g3_selection.npz")
    s = SEL_CASE
    maps, maps_t = selection_maps()
    md, mtd = maps.cuda(), maps_t.cuda()
    Rr = s["R"]
    for ns in (1, 2):
        am, kl = ops.token_stats(md, num_subjects=ns, sigma=s["sigma"])
        ref_pts = t(g[f"find_k_max_pixels_{ns}"])                              # [ns, T, 2] (row+.5, col+.5)
        ref_flat = ((ref_pts[..., 0] - 0.5) * Rr + (ref_pts[..., 1] - 0.5)).long()
        assert torch.equal(am.cpu().long(), ref_flat)
        torch.testing.assert_close(kl.cpu(), t(g[f"kl_ns{ns}"]), rtol=2e-5, atol=1e-6)
        am_t, _ = ops.token_stats(mtd, num_subjects=1, sigma=s["sigma"], want_kl=False)
        cand, sel = ops.select_tokens(kl, am_t[0], Rr, s["n_cand"], s["top_k"])
        assert torch.equal(cand.cpu(), t(g[f"top_k_gaussian_ns{ns}"]))
        assert torch.equal(sel.cpu(), t(g[f"fps_ns{ns}"]))
    assert am[0, 5].item() == 3 * Rr + 4                                        # first index wins the tie


def test_g4_losses_and_gradients(ops, golden):
    g = golden("g4_losses.npz")
    s = SEL_CASE
    maps, maps_t = selection_maps()
    sel = torch.tensor(s["sel"]).cuda()
    thetas = t(g["theta"])
    for ns in (1, 2):
        for index in (0, 1):
            a = maps.cuda().requires_grad_(True)
            b = maps_t.cuda().requires_grad_(True)
            am, _ = ops.token_stats(a, num_subjects=ns, sigma=s["sigma"], want_kl=False)
            sharp, equiv = ops.fused_losses(a, b, sel, am, thetas[index].reshape(-1).tolist(), s["sigma"], ns)
            assert abs(sharp.item() - float(g[f"sharp_ns{ns}"])) < 1e-5 * abs(float(g[f"sharp_ns{ns}"]))
            assert abs(equiv.item() - float(g[f"equiv_{index}"])) < 1e-4 * abs(float(g[f"equiv_{index}"]))
            (sharp * 3.0 + equiv * 7.0).backward()
            ga = 3.0 * t(g[f"sharp_grad_ns{ns}"]) + 7.0 * t(g[f"equiv_grad_a_{index}"])
            gb = 7.0 * t(g[f"equiv_grad_b_{index}"])
            torch.testing.assert_close(a.grad[sel].cpu(), ga, rtol=1e-4, atol=2e-5 * ga.abs().max().item())
            torch.testing.assert_close(b.grad[sel].cpu(), gb, rtol=1e-3, atol=2e-5 * gb.abs().max().item())
            mask = torch.ones(maps.shape[0], dtype=torch.bool); mask[sel.cpu()] = False
            assert a.grad.cpu()[mask].abs().max().item() == 0.0


def test_g5_subgraph_gradient_vs_reference(ops, golden):
    """Two views -> fused maps -> on-device selection -> fused losses -> d/d context (through to_k)."""
    g = golden("g5_subgraph.npz")
    e = E2E_CASE
    context = seeded((1, e["T"], e["ctx_dim"]), e["seed"] + 200).cuda().requires_grad_(True)
    maps = []
    for view in (0, 1):
        qs, ks, scales = _stack_qk(e, context, view=view)
        maps.append(ops.attn_map(qs, ks, e["heads"], scales, e["R"])[0])
    am, am_t = maps
    torch.testing.assert_close(am.detach().cpu(), t(g["map"]), **MAP_TOL)
    torch.testing.assert_close(am_t.detach().cpu(), t(g["map_t"]), **MAP_TOL)
    st, kl = ops.token_stats(am, 1, e["sigma"])
    st_t, _ = ops.token_stats(am_t, 1, e["sigma"], want_kl=False)
    cand, sel = ops.select_tokens(kl, st_t[0], e["R"], e["n_cand"], e["top_k"])
    assert torch.equal(cand.cpu(), t(g["cand"]))
    assert torch.equal(sel.cpu(), t(g["sel"]))
    theta = R.affine_matrix(11.0, 0.87, (0.13, -0.21)).reshape(-1).tolist()
    sharp, equiv = ops.fused_losses(am, am_t, sel, st, theta, e["sigma"], 1)
    assert abs(sharp.item() - float(g["sharp"])) < 1e-4 * abs(float(g["sharp"]))
    assert abs(equiv.item() - float(g["equiv"])) < 1e-3 * abs(float(g["equiv"]))
    (equiv * 1000.0 + sharp * 100.0).backward()
    ref = t(g["context_grad"])
    torch.testing.assert_close(context.grad.cpu(), ref, rtol=2e-3, atol=2e-5 * ref.abs().max().item())


def test_sd_shape_maps_invariants(ops):
    """BASELINE config-2 shapes (3x(16^2,C=1280) + 1x(32^2,C=640), T=77, R=128, 8 heads): size-independent
    properties -- probabilities sum to 1 over tokens; batch rows are independent; T=100 path."""
    gen = torch.Generator().manual_seed(3)
    for T in (77, 100):
        B = 2
        qs = [torch.randn(B, 256, 1280, generator=gen).cuda() for _ in range(3)] + [torch.randn(B, 1024, 640, generator=gen).cuda()]
        ks = [torch.randn(1, T, 1280, generator=gen).cuda() for _ in range(3)] + [torch.randn(1, T, 640, generator=gen).cuda()]
        scales = [160 ** -0.5] * 3 + [80 ** -0.5]
        M = ops.attn_map(qs, ks, 8, scales, 128)
        assert M.shape == (B, T, 128, 128)
        torch.testing.assert_close(M.sum(1), torch.ones(B, 128, 128, device="cuda"), rtol=1e-5, atol=1e-5)
        M1 = ops.attn_map([q[1:2] for q in qs], ks, 8, scales, 128)
        torch.testing.assert_close(M1[0], M[1], rtol=0, atol=0)


@pytest.mark.parametrize("B,Bk,N,H,d,T", [(2, 1, 256, 8, 160, 77), (1, 1, 100, 4, 8, 16), (2, 2, 1024, 8, 80, 77),
                                          (1, 1, 4096, 8, 40, 100), (3, 1, 64, 2, 16, 5), (1, 1, 576, 5, 64, 77),
                                          (2, 1, 96, 3, 32, 30),
                                          # token-split form (small layers: a wave per 32-token tile): 8^2 layer, ragged query block
                                          # with four tiles, two tiles, four full tiles with per-row keys
                                          (8, 1, 64, 8, 160, 77), (2, 1, 250, 4, 160, 100), (1, 1, 96, 3, 80, 40), (3, 3, 200, 2, 80, 128)])
def test_cross_attention_fwd_bwd_vs_fp64(ops, B, Bk, N, H, d, T):
    """Fused fp32-MFMA cross-attention (ptp_utils.py:493-506) against the reference formulation in fp64."""
    g = torch.Generator().manual_seed(11)
    C = H * d
    q = torch.randn(B, N, C, generator=g)
    k = torch.randn(Bk, T, C, generator=g)
    v = torch.randn(Bk, T, C, generator=g)
    w = torch.randn(B, N, C, generator=g)
    scale = d ** -0.5
    qd, kd, vd = (x.double().requires_grad_(True) for x in (q, k, v))
    qh = R.split_heads(qd, H)
    kh = R.split_heads(kd.expand(B, -1, -1), H)
    vh = R.split_heads(vd.expand(B, -1, -1), H)
    ref = R.merge_heads(torch.matmul((torch.einsum("bid,bjd->bij", qh, kh) * scale).softmax(-1), vh), H)
    (ref * w.double()).sum().backward()
    qg, kg, vg = (x.cuda().requires_grad_(True) for x in (q, k, v))
    out = ops.cross_attention(qg, kg, vg, H, scale)
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-5)
    (out * w.cuda()).sum().backward()
    for a, b in ((qg, qd), (kg, kd), (vg, vd)):
        torch.testing.assert_close(a.grad.cpu().double(), b.grad, rtol=1e-3, atol=2e-5 * b.grad.abs().max().item())


def test_many_tokens_grouped_path_matches_fp64(ops):
    """T = 300 learned tokens (reference CLI default is 500): token-group two-pass path, fwd + bwd."""
    c = STACK_CASE
    heads, Rr, T, B = c["heads"], 32, 300, 2
    g = torch.Generator().manual_seed(9)
    qs = [torch.randn(B, sl * sl, Cl, generator=g) for sl, Cl in c["layers"]]
    ks = [torch.randn(1, T, Cl, generator=g) for sl, Cl in c["layers"]]
    W = torch.randn(B, T, Rr, Rr, generator=g)
    scales = [(Cl // heads) ** -0.5 for _, Cl in c["layers"]]
    qd = [q.double().requires_grad_(True) for q in qs]
    kd = [k.double().requires_grad_(True) for k in ks]
    maps = []
    for q, k, (sl, Cl), sc in zip(qd, kd, c["layers"], scales):
        qi = q.reshape(B, sl, sl, Cl).permute(0, 3, 1, 2)
        qu = torch.nn.functional.interpolate(qi, size=(Rr, Rr), mode="bicubic", align_corners=False)
        qu = R.split_heads(qu.permute(0, 2, 3, 1).reshape(B, Rr * Rr, Cl), heads)
        kk = R.split_heads(k.expand(B, -1, -1), heads)
        p = (torch.einsum("bid,bjd->bij", qu, kk) * sc).softmax(-1)
        maps.append(p.reshape(B, heads, Rr, Rr, T).permute(0, 1, 4, 2, 3))
    Mref = torch.stack(maps, 0).mean(dim=(0, 2))
    (Mref * W.double()).sum().backward()
    qg = [q.cuda().requires_grad_(True) for q in qs]
    kg = [k.cuda().requires_grad_(True) for k in ks]
    M = ops.attn_map(qg, kg, heads, scales, Rr)
    torch.testing.assert_close(M.detach().cpu().double(), Mref.detach(), rtol=1e-3, atol=1e-6)
    torch.testing.assert_close(M.sum(1), torch.ones(B, Rr, Rr, device="cuda"), rtol=1e-5, atol=1e-5)
    (M * W.cuda()).sum().backward()
    for a, b in zip(qg + kg, qd + kd):
        torch.testing.assert_close(a.grad.cpu().double(), b.grad, rtol=2e-3, atol=2e-5 * b.grad.abs().max().item())


def _ref_maps_fp64(qd, kd, layers, heads, scales, Rr, B):
    """Reference order of operations (ptp_utils.py:513-538 + optimize.py:27-79) in fp64: upsample the
    layer input's queries, per-head softmax over tokens, mean over layers and heads -> [B,T,R,R]."""
    maps = []
    for q, k, (sl, Cl), sc in zip(qd, kd, layers, scales):
        T = k.shape[1]
        qi = q.reshape(B, sl, sl, Cl).permute(0, 3, 1, 2)
        qu = torch.nn.functional.interpolate(qi, size=(Rr, Rr), mode="bicubic", align_corners=False)
        qu = R.split_heads(qu.permute(0, 2, 3, 1).reshape(B, Rr * Rr, Cl), heads)
        kk = R.split_heads(k.expand(B, -1, -1), heads)
        p = (torch.einsum("bid,bjd->bij", qu, kk) * sc).softmax(-1)
        maps.append(p.reshape(B, heads, Rr, Rr, T).permute(0, 1, 4, 2, 3))
    return torch.stack(maps, 0).mean(dim=(0, 2))


@pytest.mark.parametrize("name,layers,heads,T,Rr", [
    # BASELINE configs[3]: SD-2.1 at 768^2 -> latents 96^2; the hooked "up" cross layers with seq <= 32^2 are
    # the three 24^2 layers of up_blocks.1 (C=1280, 20 heads of 64); fractional bicubic ratio on purpose
    ("sd21_768", [(24, 1280)] * 3, 20, 77, 100),
    # BASELINE configs[0]: SD-1.5 at 256^2 -> latents 32^2: 8^2 x3 (C=1280) + 16^2 (C=1280), 8 heads
    ("sd15_256", [(8, 1280)] * 3 + [(16, 1280)], 8, 77, 64),
    # BASELINE configs[4]: SDXL at 1024^2 -> latents 128^2; hooked layers are 32^2, C=1280, 20 heads of 64
    # (two of the four layers, to keep the fp64 CPU side at a few seconds)
    ("sdxl_1024", [(32, 1280)] * 2, 20, 77, 128),
])
def test_other_baseline_config_shapes_vs_fp64(ops, name, layers, heads, T, Rr):
    """The attention-map path at the shapes of the other BASELINE configs (parity-test cases, not bench
    lines): forward map rtol 1e-3 and the q/k gradients against fp64 autograd of the reference order."""
    B = 2
    g = torch.Generator().manual_seed(17)
    qs = [torch.randn(B, sl * sl, Cl, generator=g) for sl, Cl in layers]
    ks = [torch.randn(1, T, Cl, generator=g) for sl, Cl in layers]
    W = torch.randn(B, T, Rr, Rr, generator=g)
    scales = [(Cl // heads) ** -0.5 for _, Cl in layers]
    qd = [q.double().requires_grad_(True) for q in qs]
    kd = [k.double().requires_grad_(True) for k in ks]
    Mref = _ref_maps_fp64(qd, kd, layers, heads, scales, Rr, B)
    (Mref * W.double()).sum().backward()
    qg = [q.cuda().requires_grad_(True) for q in qs]
    kg = [k.cuda().requires_grad_(True) for k in ks]
    M = ops.attn_map(qg, kg, heads, scales, Rr)
    torch.testing.assert_close(M.detach().cpu().double(), Mref.detach(), rtol=1e-3, atol=1e-6)
    (M * W.cuda()).sum().backward()
    for a, b in zip(qg + kg, qd + kd):
        torch.testing.assert_close(a.grad.cpu().double(), b.grad, rtol=2e-3, atol=2e-5 * b.grad.abs().max().item())


@pytest.mark.parametrize("N,C,G,H,W,silu,with_off", [(2, 64, 32, 16, 16, True, True), (8, 320, 32, 64, 64, True, True),
                                                      (1, 128, 32, 40, 24, False, False), (3, 32, 32, 6, 6, True, False),
                                                      (2, 512, 32, 128, 128, True, False),
                                                      # one-pass forms (row in registers): 2 / 4 / 8-10 / 16 float4 per thread; the
                                                      # last one's backward (61 440 elements per row) takes the two-kernel form
                                                      (2, 2560, 32, 8, 8, False, True), (2, 1280, 32, 16, 16, True, False),
                                                      (2, 640, 32, 32, 32, True, True), (2, 1920, 32, 32, 32, True, True)])
def test_fused_group_norm_silu_fwd_bwd(ops, N, C, G, H, W, silu, with_off):
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(N, C, H, W, generator=g) * 2 + 0.7)
    off = torch.randn(N, C, generator=g) if with_off else None
    norm = torch.nn.GroupNorm(G, C, eps=1e-5)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g)); norm.bias.copy_(torch.randn(C, generator=g))
    w = torch.randn(N, C, H, W, generator=g)
    xd = x.double().requires_grad_(True)
    nd = torch.nn.GroupNorm(G, C, eps=1e-5).double()
    nd.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    z = nd(xd + (off.double()[:, :, None, None] if with_off else 0))
    ref = torch.nn.functional.silu(z) if silu else z
    (ref * w.double()).sum().backward()
    norm = norm.cuda()
    xg = x.cuda().requires_grad_(True)
    y = ops.group_norm_silu(xg, norm, off=off.cuda() if with_off else None, silu=silu)
    torch.testing.assert_close(y.detach().cpu().double(), ref.detach(), rtol=2e-5, atol=2e-5)
    (y * w.cuda()).sum().backward()
    torch.testing.assert_close(xg.grad.cpu().double(), xd.grad, rtol=1e-4, atol=2e-5 * xd.grad.abs().max().item())


@pytest.mark.parametrize("B,N,H,d", [(2, 1024, 8, 80), (1, 4096, 8, 40), (2, 200, 4, 8), (1, 256, 8, 160), (3, 64, 2, 16),
                                     (1, 130, 4, 40), (1, 576, 5, 64), (2, 100, 2, 32), (2, 1100, 2, 40), (1, 1281, 3, 40), (1, 1100, 2, 64), (2, 1024, 3, 64),
                                     # few workgroups: range splits of the backward (partials + fixed-order sum): four / two splits of the
                                     # two-kernel form, a ragged last tile, a split that owns no tile, the fused form's two query halves
                                     (2, 256, 8, 160), (4, 256, 8, 160), (3, 200, 2, 160), (2, 288, 8, 160), (1, 1100, 8, 80)])
def test_flash_self_attention_fwd_bwd_vs_fp64(ops, B, N, H, d):
    """Flash-style fp32-MFMA self-attention against the materialised reference formulation in fp64."""
    g = torch.Generator().manual_seed(13)
    C = H * d
    q, k, v, w = (torch.randn(B, N, C, generator=g) for _ in range(4))
    scale = d ** -0.5
    qd, kd, vd = (x.double().requires_grad_(True) for x in (q, k, v))
    qh, kh, vh = R.split_heads(qd, H), R.split_heads(kd, H), R.split_heads(vd, H)
    ref = R.merge_heads(torch.matmul((torch.einsum("bid,bjd->bij", qh, kh) * scale).softmax(-1), vh), H)
    (ref * w.double()).sum().backward()
    qg, kg, vg = (x.cuda().requires_grad_(True) for x in (q, k, v))
    out = ops.self_attention(qg, kg, vg, H, scale)
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-5)
    (out * w.cuda()).sum().backward()
    for a, b in ((qg, qd), (kg, kd), (vg, vd)):
        torch.testing.assert_close(a.grad.cpu().double(), b.grad, rtol=1e-3, atol=2e-5 * b.grad.abs().max().item())


def test_sd_shape_backward_is_deterministic_and_finite(ops):
    """BASELINE config-2 shapes: the atomics-free backward is bit-reproducible run to run; gradients are finite
    and the gradient of sum_t M (== 1 everywhere) is zero (softmax rows sum to one)."""
    gen = torch.Generator().manual_seed(8)
    B, T = 2, 77
    qs = [torch.randn(B, 256, 1280, generator=gen).cuda().requires_grad_(True) for _ in range(3)] + \
         [torch.randn(B, 1024, 640, generator=gen).cuda().requires_grad_(True)]
    ks = [torch.randn(1, T, 1280, generator=gen).cuda().requires_grad_(True) for _ in range(3)] + \
         [torch.randn(1, T, 640, generator=gen).cuda().requires_grad_(True)]
    scales = [160 ** -0.5] * 3 + [80 ** -0.5]
    W = torch.randn(B, T, 128, 128, generator=gen).cuda()
    grads = []
    for _ in range(2):
        M = ops.attn_map(qs, ks, 8, scales, 128)
        g = torch.autograd.grad((M * W).sum(), qs + ks)
        grads.append([x.clone() for x in g])
    for a, b in zip(*grads):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    M = ops.attn_map(qs, ks, 8, scales, 128)
    g1 = torch.autograd.grad(M.sum(), qs + ks)                  # d/dq sum_t softmax_t == 0
    for a, ref in zip(g1, grads[0]):
        assert a.abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("B,ci,co,H,W,bias", [(2, 32, 32, 8, 8, True), (1, 64, 96, 7, 5, False), (3, 32, 160, 6, 10, True),
                                              (8, 128, 128, 64, 64, True), (2, 320, 320, 16, 16, False),
                                              (1, 96, 64, 33, 17, True), (3, 64, 96, 12, 20, True), (8, 1280, 640, 8, 8, False)])
def test_winograd_conv3x3_fwd_bwd_vs_fp64(ops, B, ci, co, H, W, bias):
    """3x3/s1/p1 convolution of the frozen blocks (Winograd F(2x2,3x3) on fp32 MFMA) against fp64 conv2d: forward
    (+bias), and the input gradient through the same kernel with the rotated/transposed filter.  Both workgroup
    shapes; odd sizes exercise partial tiles, ragged tile blocks and idle channel waves."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    b = torch.randn(co, generator=g) if bias else None
    gy = torch.randn(B, co, H, W, generator=g)
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xd, w.double(), None if b is None else b.double(), padding=1)
    (ref * gy.double()).sum().backward()
    xg, wg = x.cuda().requires_grad_(True), w.cuda()
    bg = None if b is None else b.cuda()
    assert ops.conv3x3_supported(xg.shape, wg.shape)
    y = ops.conv3x3(xg, wg, bg)                            # F(4x4,3x3) or F(2x2,3x3) by shape
    tol = 2e-5 * ref.abs().max().item()
    tol4 = 6e-5 * ref.abs().max().item()                   # F(4x4,3x3): ~1 digit less than F(2x2,3x3), see skp.h
    torch.testing.assert_close(y.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=tol4)
    if H % 4 == 0 and W % 4 == 0 and ci % 16 == 0 and co % 16 == 0:
        y4 = ops._conv3x3_f4_raw(xg.detach(), ops._wino4_filters(wg, False), bg, co)
        torch.testing.assert_close(y4.cpu().double(), ref.detach(), rtol=1e-4, atol=tol4)
        y4n = ops._conv3x3_f4_raw(xg.detach(), ops._wino4_filters(wg, False), bg, co, split=False)
        torch.testing.assert_close(y4n.cpu().double(), ref.detach(), rtol=1e-4, atol=tol4)
        dx4 = ops._conv3x3_f4_raw(gy.cuda(), ops._wino4_filters(wg, True), None, ci)
        torch.testing.assert_close(dx4.cpu().double(), xd.grad, rtol=1e-4, atol=6e-5 * xd.grad.abs().max().item())
    for variant in (1, 2):
        yv = ops._conv3x3_raw(xg.detach(), ops._wino_filters(wg, False), bg, co, variant)
        torch.testing.assert_close(yv.cpu().double(), ref.detach(), rtol=1e-4, atol=tol)
    dx = ops._conv3x3_raw(gy.cuda(), ops._wino_filters(wg, True), None, ci)
    torch.testing.assert_close(dx.cpu().double(), xd.grad, rtol=1e-4, atol=2e-5 * xd.grad.abs().max().item())
    (y * gy.cuda()).sum().backward()                       # autograd path (kernel or library by size)
    torch.testing.assert_close(xg.grad.cpu().double(), xd.grad, rtol=1e-4, atol=6e-5 * xd.grad.abs().max().item())
    # determinism
    assert torch.equal(ops.conv3x3(xg.detach(), wg, bg), y.detach())
    # residual (ResnetBlock2D shortcut) added in the epilogue; its gradient is dy
    res = torch.randn(B, co, H, W, generator=g).cuda().requires_grad_(True)
    yr = ops.conv3x3(xg.detach(), wg, bg, res)
    torch.testing.assert_close(yr.detach(), y.detach() + res.detach(), rtol=1e-5, atol=1e-5)
    (yr * gy.cuda()).sum().backward()
    assert torch.equal(res.grad, gy.cuda())


def test_geglu_and_layout_kernels(ops):
    """Fused GEGLU (fwd/bwd) vs torch in fp64; NCHW<->token transposes (+ residual) exact, with their gradients."""
    g = torch.Generator().manual_seed(23)
    p = torch.randn(3, 50, 2 * 72, generator=g)
    w = torch.randn(3, 50, 72, generator=g)
    pd = p.double().requires_grad_(True)
    h, gate = pd.chunk(2, dim=-1)
    ref = h * torch.nn.functional.gelu(gate)
    (ref * w.double()).sum().backward()
    pg = p.cuda().requires_grad_(True)
    y = ops.geglu(pg)
    torch.testing.assert_close(y.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=1e-6)
    (y * w.cuda()).sum().backward()
    torch.testing.assert_close(pg.grad.cpu().double(), pd.grad, rtol=1e-5, atol=1e-6)
    for (B, C, H, W) in [(2, 64, 8, 8), (3, 100, 6, 10), (1, 320, 64, 64)]:
        x = torch.randn(B, C, H, W, generator=g).cuda().requires_grad_(True)
        r = torch.randn(B, C, H, W, generator=g).cuda().requires_grad_(True)
        t = ops.nchw_to_tokens(x)
        assert torch.equal(t, x.permute(0, 2, 3, 1).reshape(B, H * W, C))
        back = ops.tokens_to_nchw_add(t, r)
        assert torch.equal(back, x + r)
        gy = torch.randn(B, C, H, W, generator=g).cuda()
        (back * gy).sum().backward()
        assert torch.equal(x.grad, gy) and torch.equal(r.grad, gy)


def test_winograd_conv3x3_batch_chunking_over_2gib(ops):
    """Tensors above the kernels' 2 GiB per-launch addressing limit are launched in batch chunks: same result as the
    per-image launches."""
    g = torch.Generator().manual_seed(29)
    B, C, H = 5, 128, 1024                                  # 5 x 128 x 1024^2 x 4 B = 2.5 GiB
    x = torch.randn(B, C, H, H, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).cuda()
    b = torch.randn(C, generator=g).cuda()
    y = ops.conv3x3(x, w, b)
    for i in (0, 3, 4):
        yi = ops.conv3x3(x[i:i + 1].contiguous(), w, b)
        assert torch.equal(y[i:i + 1], yi)


@pytest.mark.gpu
@pytest.mark.parametrize("B,ci,co,H,W,bias,res", [(8, 1280, 1280, 8, 8, True, True), (8, 640, 1280, 16, 16, False, False),
                                                   (8, 2560, 1280, 8, 8, True, False), (2, 256, 64, 16, 16, True, True),
                                                   (3, 320, 128, 8, 12, False, True), (1, 272, 192, 4, 4, True, False),
                                                   (2, 1280, 1280, 8, 8, True, True),        # config 3's per-rank rows: 8 tiles -> the one-block form
                                                   (4, 1280, 1280, 8, 8, False, True), (1, 640, 1280, 16, 16, True, False)])   # exactly 16 tiles
def test_raw_filter_winograd_small_layers_vs_fp64(ops, B, ci, co, H, W, bias, res):
    """Small-spatial form of the 3x3 convolution (raw taps, G g G^T applied in the lanes, input transform in the workspace:
    skp_conv3x3_f4r_f32) against fp64 conv2d: forward with bias / residual, the input gradient through the same kernel with
    the rotated / transposed taps, K-split and unsplit plans, ragged tile blocks (12 and 1 tiles of 32) and a channel count
    that is not a multiple of 64 on the reduction side.  Same tolerance as the other F(4x4,3x3) kernels; bit-reproducible."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    b = torch.randn(co, generator=g) if bias else None
    r = torch.randn(B, co, H, W, generator=g) if res else None
    gy = torch.randn(B, co, H, W, generator=g)
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xd, w.double(), None if b is None else b.double(), padding=1)
    if r is not None:
        ref = ref + r.double()
    (ref * gy.double()).sum().backward()
    lib = ops.N.lib()
    assert lib.skp_conv3x3_f4r_workspace(B, ci, co, H, W) > 0                    # a layout the kernel runs
    assert lib.skp_conv3x3_f4r_ok(B, ci, co, H, W) == (1 if min(ci, co) >= 1280 else 0)     # ... and where it is routed to
    xg, wg = x.cuda(), w.cuda()
    bg, rg = (None if b is None else b.cuda()), (None if r is None else r.cuda())
    y = ops._conv3x3_f4r_raw(xg, ops._wino4r_filters(wg, False), bg, co, residual=rg)
    torch.testing.assert_close(y.cpu().double(), ref.detach(), rtol=1e-4, atol=6e-5 * ref.abs().max().item())
    assert torch.equal(ops._conv3x3_f4r_raw(xg, ops._wino4r_filters(wg, False), bg, co, residual=rg), y)
    for split in ("1", "2"):                                     # the unsplit kernel form (bias / residual in its own epilogue) and a forced K split
        ops.N.tune("wino_split", int(split))
        try:
            ys = ops._conv3x3_f4r_raw(xg, ops._wino4r_filters(wg, False), bg, co, residual=rg)
        finally:
            ops.N.tune("wino_split", 0)
        torch.testing.assert_close(ys.cpu().double(), ref.detach(), rtol=1e-4, atol=6e-5 * ref.abs().max().item())
    if ci % 64 == 0:                                             # the backward-data launch swaps the channel roles
        dx = ops._conv3x3_f4r_raw(gy.cuda(), ops._wino4r_filters(wg, True), None, ci)
        torch.testing.assert_close(dx.cpu().double(), xd.grad, rtol=1e-4, atol=6e-5 * xd.grad.abs().max().item())
        # and through autograd (Conv3x3Fn routes both directions by shape)
        xq = xg.clone().requires_grad_(True)
        routed = lib.skp_conv3x3_f4r_ok(B, ci, co, H, W) == 1 and lib.skp_conv3x3_f4r_ok(B, co, ci, H, W) == 1
        yq = ops.conv3x3_auto(xq, wg, bg, rg) if routed else None
        if yq is not None:
            torch.testing.assert_close(yq.detach(), y, rtol=0, atol=0)
            (yq * gy.cuda()).sum().backward()
            torch.testing.assert_close(xq.grad, dx, rtol=0, atol=0)
