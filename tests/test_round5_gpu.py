"""GPU tests added in round 5.

* `python bench.py --gpus 2` WITHOUT a launcher around it (the N > 1 line must not depend on who wraps the command);
* RCCL itself on the box: `init_process_group("nccl", world_size=1)`, the embedding-gradient all-reduce ordered against the
  step's HIP kernels, 1000 calls timed (tools/rccl_smoke.py);
* the protocol bench of BASELINE config 2 (`optimize_embedding` as the reference runs it) on a short, reduced-width run:
  host-resident folder data through the one-batch-ahead loader gives the SAME embedding as the synchronous loop.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_self_launch_two_ranks_one_gpu():
    """No torchrun, no WORLD_SIZE: bench.py re-runs itself under torch.distributed.run (both ranks on cuda:0, gloo standing in
    for RCCL on the one-GPU box), rank 0 prints the one JSON line, exit status 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SKP_DIST_BACKEND="gloo", SKP_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--image-size", "128", "--res", "32",
           "--tokens", "16", "--steps", "2", "--warmup", "1", "--cpu-baseline", "off", "--kernel-iters", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = _json_lines(out)
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0
    assert d["collective_check"]["ok"] and d["collective_check"]["ranks_seen"] == 2
    assert d["collective_check"]["embedding_identical_on_all_ranks"] is True
    assert "torch.distributed.run" in out.stderr


def test_rccl_single_rank_allreduce_on_step_stream():
    """RCCL on the box (a13 / (e)): communicator init, the [1,77,768] gradient all-reduced 1000x on the step's stream, the
    value, the ordering against the step's kernels, us per call."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py"), "--calls", "1000"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    lines = _json_lines(out)
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    d = lines[0]
    print("RCCL single-rank:", json.dumps(d))
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["elements"] == 77 * 768
    assert d["value_ok"] and d["ordered_against_step_kernels"]
    assert 0 < d["us_per_call_stream"] < 500


def test_optimize_embedding_host_folder_prefetch_is_order_identical(tmp_path):
    """`optimize_embedding` on a HOST-resident folder of PNG files (`dataset_name="custom"`, reference
    datasets/custom_images.py) through the one-group-ahead loader (decode threads + pinned staging + async H2D) against
    the synchronous loop.  The loader itself is checked exactly: the same indices and bit-identical pixels, group by group,
    on the GPU.  The two optimisation runs then see identical inputs; their embeddings agree to the run-to-run noise of the
    step (library GEMM / convolution reductions are not bit-reproducible between runs), measured here by repeating the
    synchronous run; the step callback fires once per optimizer step."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from protocol_bench import write_png_folder
    from stablekeypoints_amd.custom_images import CustomDataset
    from stablekeypoints_amd.optimize import GroupLoader, _group_indices, default_args, optimize_embedding
    from stablekeypoints_amd.optimize_token import load_ldm
    folder = write_png_folder(str(tmp_path / "imgs"), 6, 128, seed=4)

    def mk(workers, steps=4):
        return default_args(dataset_name="custom", dataset_loc=folder, num_steps=steps, batch_size=2, num_tokens=16,
                            feature_upsample_res=32, furthest_point_num_samples=8, top_k=4, image_size=128, device="cuda",
                            log_interval=0, loader_workers=workers, seed=5)
    ds = CustomDataset(folder, 128)
    a = mk(0, steps=7)
    sync = list(GroupLoader(ds, _group_indices(a, len(ds), None, 2, 2, 0, 1), "cuda", workers=0))
    pre = GroupLoader(ds, _group_indices(a, len(ds), None, 2, 2, 0, 1), "cuda", workers=3)
    assert pre.assembler is not None                                # the threaded mode really is on
    ahead = list(pre)
    pre.close()
    assert len(sync) == len(ahead) == 7
    for (it0, idx0, im0), (it1, idx1, im1) in zip(sync, ahead):
        assert it0 == it1 and idx0 == idx1 and im1.is_cuda and torch.equal(im0.cuda(), im1)
    ldm, controllers, n = load_ldm("cuda", "tiny", feature_upsample_res=32)
    trajs = []
    for workers in (0, 0, 3):
        torch.manual_seed(11)                                       # the affine / noise draws of the loop
        torch.cuda.manual_seed(11)
        traj, calls = [], []
        ctx0 = torch.randn(1, 16, 768, generator=torch.Generator().manual_seed(2)) * 3.0
        out = optimize_embedding(ldm, mk(workers), controllers, n, context=ctx0.clone(), trajectory_out=traj, step_callback=calls.append)
        assert calls == [0, 1, 2, 3] and torch.equal(out, traj[-1])
        trajs.append(torch.cat(traj).cpu())
    lr = 5e-3
    noise = (trajs[0] - trajs[1]).abs().max().item()
    diff = (trajs[0] - trajs[2]).abs().max().item()
    print(f"run-to-run noise of the synchronous loop {noise / lr:.3e} lr; prefetch vs synchronous {diff / lr:.3e} lr")
    assert (trajs[0][0] - trajs[0][-1]).abs().max().item() > lr    # the embedding moved
    assert diff <= max(4 * noise, 0.02 * lr)


# ---------------------------------------------------------------------------------------------------------------------
# flash attention forward on the bf16 matrix cores with three-term operand splits (csrc/skp_flash_attn_s.hip)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Bk,H,N,Nk,d", [
    (2, 2, 8, 1024, 1024, 40),        # self-attention, the 64^2-layer head size (row sums through the spare row of O^T)
    (1, 1, 4, 300, 200, 40),          # ragged query block and ragged last key tile
    (3, 1, 2, 64, 500, 40),           # one k / v shared by all rows (cross-attention with many tokens)
    (2, 2, 8, 1024, 1024, 80),        # d % 16 == 0: VALU row sums, the full 160 KB of LDS
    (1, 1, 3, 100, 77, 80),           # a single ragged tile
    (4, 4, 8, 2048, 2048, 40),        # >= 256 workgroups of 256 queries: 32 queries per wave (the shapes above run 16 per wave)
    (4, 4, 8, 2048, 2048, 80),
])
def test_split_bf16_flash_attention_forward_vs_fp64(B, Bk, H, N, Nk, d):
    """skp_flash_attn_fwd_split_f32 against fp64 softmax(scale q k^T) v and against the fp32-instruction kernel on the same
    inputs: out and lse; error at most 1.5x the fp32 kernel's (+ 1e-7 of the scale); a spiked key (one logit far above the
    rest in the middle of the key axis) exercises the running-max rescale; deterministic."""
    from stablekeypoints_amd import ops
    g = torch.Generator().manual_seed(9)
    C = H * d
    q = torch.randn(B, N, C, generator=g)
    k = torch.randn(Bk, Nk, C, generator=g)
    v = torch.randn(Bk, Nk, C, generator=g)
    k[:, Nk // 2] *= 6.0                                                          # the spike
    scale = d ** -0.5
    qd = q.double().reshape(B, N, H, d).permute(0, 2, 1, 3)
    kd = k.double().reshape(Bk, Nk, H, d).permute(0, 2, 1, 3).expand(B, H, Nk, d)
    vd = v.double().reshape(Bk, Nk, H, d).permute(0, 2, 1, 3).expand(B, H, Nk, d)
    sc = qd @ kd.transpose(-1, -2) * scale
    ref = (sc.softmax(-1) @ vd).permute(0, 2, 1, 3).reshape(B, N, C)
    ref_lse = torch.logsumexp(sc, dim=-1)
    assert ops.N.lib().skp_flash_attn_fwd_split_ok(B, Bk, H, N, Nk, d) == 1
    qg, kg, vg = q.cuda(), k.cuda(), v.cuda()
    out, lse = ops.flash_attn_fwd_split(qg, kg, vg, H, scale)
    o32 = torch.empty_like(qg); l32 = torch.empty(B, H, N, device="cuda")
    ops.N.check(ops.N.lib().skp_flash_attn_fwd_f32(qg.data_ptr(), kg.data_ptr(), vg.data_ptr(), o32.data_ptr(), l32.data_ptr(),
                                                   B, Bk, H, N, Nk, d, float(scale), ops._stream()), "fp32 flash forward")
    sref = ref.abs().max().item()
    e_s, e_32 = (out.cpu().double() - ref).abs().max().item() / sref, (o32.cpu().double() - ref).abs().max().item() / sref
    el_s, el_32 = (lse.cpu().double() - ref_lse).abs().max().item(), (l32.cpu().double() - ref_lse).abs().max().item()
    print(f"d={d} N={N} Nk={Nk}: out err split {e_s:.2e} fp32 {e_32:.2e} (ratio {e_s / e_32:.2f}); lse err split {el_s:.2e} fp32 {el_32:.2e}")
    assert e_s <= 1.5 * e_32 + 1e-7 and e_s < (2e-6 if Nk <= 1024 else 1e-5)      # (the spiked key's error grows with the key count: 5e-6 at 2048, fp32 kernel 1.4e-5)
    assert el_s <= 1.5 * el_32 + 2e-6
    out2, lse2 = ops.flash_attn_fwd_split(qg, kg, vg, H, scale)
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    assert ops.N.lib().skp_flash_attn_fwd_split_ok(1, 1, 8, 64, 64, 64) == 0      # head sizes without a split kernel stay on fp32


@pytest.mark.parametrize("B,H,N", [(2, 8, 1024), (1, 3, 300), (1, 2, 77)])
def test_split_bf16_flash_attention_backward_vs_fp64(B, H, N):
    """skp_flash_attn_bwd_split_f32 (d = 40 self-attention: dQ kernel + dK / dV kernel on three-term bf16 tuples) against fp64
    autograd through softmax(scale q k^T) v and against the fp32-instruction backward on the same inputs (ragged query / key
    tiles included); each gradient's error at most 1.5x the fp32 kernels' (+ 1e-7 of its scale); deterministic."""
    from stablekeypoints_amd import ops
    d = 40
    g = torch.Generator().manual_seed(19)
    C = H * d
    q, k, v, go = (torch.randn(B, N, C, generator=g) for _ in range(4))
    k[:, N // 3] *= 5.0
    scale = d ** -0.5
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    def heads(t):
        return t.reshape(B, N, H, d).permute(0, 2, 1, 3)
    ref = ((heads(qd) @ heads(kd).transpose(-1, -2) * scale).softmax(-1) @ heads(vd)).permute(0, 2, 1, 3).reshape(B, N, C)
    (ref * go.double()).sum().backward()
    assert ops.N.lib().skp_flash_attn_bwd_split_ok(B, B, H, N, N, d) == 1
    qg, kg, vg, gg = q.cuda(), k.cuda(), v.cuda(), go.cuda()
    out, lse = ops.flash_attn_fwd_split(qg, kg, vg, H, scale)
    dq, dk, dv = ops.flash_attn_bwd_split(qg, kg, vg, out, gg, lse, H, scale)
    q32 = qg.clone().requires_grad_(True); k32 = kg.clone().requires_grad_(True); v32 = vg.clone().requires_grad_(True)
    prev, ops.FLASH_SPLIT = ops.FLASH_SPLIT, False
    try:
        (ops.self_attention(q32, k32, v32, H, scale) * gg).sum().backward()
    finally:
        ops.FLASH_SPLIT = prev
    for name, got, g32, refg in (("dq", dq, q32.grad, qd.grad), ("dk", dk, k32.grad, kd.grad), ("dv", dv, v32.grad, vd.grad)):
        sc = refg.abs().max().item()
        e_s, e_32 = (got.cpu().double() - refg).abs().max().item() / sc, (g32.cpu().double() - refg).abs().max().item() / sc
        print(f"N={N} {name}: split {e_s:.2e}  fp32 kernels {e_32:.2e}  ratio {e_s / e_32:.2f}")
        assert e_s <= 1.5 * e_32 + 1e-7 and e_s < 5e-6
    dq2, dk2, dv2 = ops.flash_attn_bwd_split(qg, kg, vg, out, gg, lse, H, scale)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
    # through autograd with the switch on: the same kernels
    prev, ops.FLASH_SPLIT, pmin = ops.FLASH_SPLIT, True, ops.FLASH_SPLIT_MIN_KEYS
    ops.FLASH_SPLIT_MIN_KEYS = 1
    try:
        qa = qg.clone().requires_grad_(True); ka = kg.clone().requires_grad_(True); va = vg.clone().requires_grad_(True)
        (ops.self_attention(qa, ka, va, H, scale) * gg).sum().backward()
    finally:
        ops.FLASH_SPLIT, ops.FLASH_SPLIT_MIN_KEYS = prev, pmin
    assert torch.equal(qa.grad, dq) and torch.equal(ka.grad, dk) and torch.equal(va.grad, dv)


@pytest.mark.parametrize("B,H,N,d,split", [(2, 8, 1024, 40, True), (2, 8, 1024, 40, False), (2, 8, 1024, 80, True), (2, 8, 256, 80, True),
                                           (3, 8, 256, 160, True), (1, 8, 64, 160, True), (1, 3, 300, 40, True)])
def test_self_attention_block_stacked_projections_vs_fp64(B, H, N, d, split, monkeypatch):
    """q | k | v as one batched GEMM + the flash backward writing dq | dk | dv as column bands of one [B*N, 3C] buffer
    (skp_flash_attn_bwd_ld_f32; at >= 1024 keys and d = 40 its split-bf16 form skp_flash_attn_bwd_split_ld_f32, the route of
    record since round 6; `split` False: the fp32-instruction kernels at the same shape) + ONE input-gradient GEMM: output and dx
    against an fp64 reference of the block (ptp_utils.py:513-520, 493-506) and against the three-projection composition it replaces."""
    from stablekeypoints_amd import ops
    monkeypatch.setattr(ops, "FLASH_SPLIT", split)
    g = torch.Generator().manual_seed(5)
    C = H * d
    x64 = torch.randn(B, N, C, generator=g, dtype=torch.float64)
    w64 = [torch.randn(C, C, generator=g, dtype=torch.float64) * C ** -0.5 for _ in range(3)]
    go64 = torch.randn(B, N, C, generator=g, dtype=torch.float64)
    scale = d ** -0.5

    def ref(x, ws):
        q, k, v = (torch.nn.functional.linear(x, w) for w in ws)
        sp = lambda t: t.reshape(B, N, H, d).permute(0, 2, 1, 3)
        p = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * scale, dim=-1)
        return (p @ sp(v)).permute(0, 2, 1, 3).reshape(B, N, C)

    xr = x64.clone().requires_grad_(True)
    o64 = ref(xr, w64)
    (dx64,) = torch.autograd.grad(o64, xr, go64)
    x = x64.float().cuda().requires_grad_(True)
    ws = [w.float().cuda() for w in w64]
    go = go64.float().cuda()
    out = ops.self_attention_block(x, *ws, H, scale)
    assert out.grad_fn is not None and type(out.grad_fn).__name__.startswith("SelfAttnQKVFn")
    (dx,) = torch.autograd.grad(out, x, go)
    x2 = x.detach().clone().requires_grad_(True)
    q, k, v = (torch.nn.functional.linear(x2, w) for w in ws)
    out2 = ops.self_attention(q, k, v, H, scale)
    (dx2,) = torch.autograd.grad(out2, x2, go)
    eo, eo2 = (out.double().cpu() - o64).abs().max().item(), (out2.double().cpu() - o64).abs().max().item()
    ed, ed2 = (dx.double().cpu() - dx64).abs().max().item(), (dx2.double().cpu() - dx64).abs().max().item()
    so, sd = o64.abs().max().item(), dx64.abs().max().item()
    print(f"B={B} H={H} N={N} d={d}: out {eo / so:.2e} (three GEMMs {eo2 / so:.2e})  dx {ed / sd:.2e} (three GEMMs {ed2 / sd:.2e})")
    assert eo <= 2.0 * eo2 + 1e-6 * so and ed <= 2.0 * ed2 + 1e-6 * sd
    assert eo < 2e-5 * so and ed < 2e-5 * sd
    # the strided entry against the dense one, bit for bit (same kernels, different row stride)
    qd, kd, vd = (t.detach().contiguous() for t in (q, k, v))
    o, lse = out2.detach(), None
    fa = ops.FlashAttnFn
    ctx_out = torch.empty_like(qd)
    lse = torch.empty(B, H, N, device="cuda")
    ops.N.check(ops.N.lib().skp_flash_attn_fwd_f32(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ctx_out.data_ptr(), lse.data_ptr(),
                                                   B, B, H, N, N, d, float(scale), ops._stream()), "fwd")
    nb = ops.N.lib().skp_flash_attn_bwd_workspace(B, B, H, N, N, d)
    wsb = torch.empty(nb // 4, device="cuda")
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(qd), torch.empty_like(qd)
    ops.N.check(ops.N.lib().skp_flash_attn_bwd_f32(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ctx_out.data_ptr(), go.data_ptr(),
                                                   lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), wsb.data_ptr(),
                                                   B, B, H, N, N, d, float(scale), ops._stream()), "bwd")
    d3 = torch.full((B * N, 3 * C + 8), float("nan"), device="cuda")
    p = d3.data_ptr()
    ops.N.check(ops.N.lib().skp_flash_attn_bwd_ld_f32(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ctx_out.data_ptr(), go.data_ptr(),
                                                      lse.data_ptr(), p, p + 4 * C, p + 8 * C, wsb.data_ptr(), B, B, H, N, N, d,
                                                      float(scale), 3 * C + 8, ops._stream()), "bwd_ld")
    assert torch.equal(d3[:, :C], dq.view(-1, C)) and torch.equal(d3[:, C:2 * C], dk.view(-1, C)) and torch.equal(d3[:, 2 * C:3 * C], dv.view(-1, C))
    assert torch.isnan(d3[:, 3 * C:]).all()                      # nothing written outside the three bands
    assert ops.N.lib().skp_flash_attn_bwd_ld_f32(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ctx_out.data_ptr(), go.data_ptr(),
                                                 lse.data_ptr(), p, p, p, wsb.data_ptr(), B, B, H, N, N, d, float(scale), C - 4,
                                                 ops._stream()) != 0


def test_batched_context_projections_equal_per_layer_projections(monkeypatch):
    """k / v of every cross-attention layer from the ONE shared context row (one batched GEMM per layer width, Bk = 1 through
    the attention and map kernels) against the per-layer projections of the expanded context (ptp_utils.py:513-520): same
    loss, same embedding gradient; layers beyond the early exit are left out from the second forward on."""
    from test_e2e_gpu import _setup
    from oracle import ref_path as R
    from stablekeypoints_amd import ptp_utils
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import group_step
    ldm, controllers, cpu, images, ctx, noise, args = _setup()
    n = images.shape[0]
    dev, controller = next(iter(controllers.items()))
    thetas = torch.cat([R.affine_matrix(11.0, 0.87, (0.13, -0.21)), R.affine_matrix(-9.0, 0.93, (-0.2, 0.1))])
    tr = RandomAffineWithInverse()
    out = {}
    for mode in (False, True, True):
        monkeypatch.setattr(ptp_utils, "CTX_KV_BATCHED", mode)
        c = ctx.clone().cuda().requires_grad_(True)
        loss, eq, sh = group_step(ldm, images, c, args, controller, tr, denom=n, noise=noise.cuda(), thetas=thetas)
        out.setdefault(mode, []).append((loss.item(), eq.item(), sh.item(), c.grad.clone()))
    plan = ldm.unet._skp_ctx_plan
    assert 0 < len(plan["used"][True]) < len(plan["mods"])         # the early exit leaves the last up-block layers out
    ref = out[False][0]
    for got in out[True]:                                          # first forward (all layers projected) and second (used ones only)
        assert abs(got[0] - ref[0]) <= 1e-5 * abs(ref[0]) and abs(got[1] - ref[1]) <= 1e-5 * abs(ref[1]) + 1e-9
        torch.testing.assert_close(got[3], ref[3], rtol=1e-4, atol=1e-5 * ref[3].abs().max().item())
    assert not ptp_utils._CTX_KV                                   # nothing of a finished forward stays behind
    # a FULL forward afterwards has its own record (all layers), and leaves the early-exit record alone
    with torch.no_grad():
        both = torch.cat([images.cuda(), tr(images.cuda(), theta=thetas)])
        cg = ctx.clone().cuda()
        outs = []
        for _ in range(2):
            _, pred = ptp_utils.find_pred_noise(ldm, both, cg, device=dev, noise=noise.cuda(), early_exit=False, controllers=controllers)
            controller.reset()
            outs.append(pred)
        monkeypatch.setattr(ptp_utils, "CTX_KV_BATCHED", False)
        _, pred_ref = ptp_utils.find_pred_noise(ldm, both, cg, device=dev, noise=noise.cuda(), early_exit=False, controllers=controllers)
        controller.reset()
    assert len(plan["used"][False]) == len(plan["mods"]) and len(plan["used"][True]) < len(plan["mods"])
    for p_ in outs:
        torch.testing.assert_close(p_, pred_ref, rtol=1e-4, atol=1e-5 * pred_ref.abs().max().item())
    # a layer whose projection is unfrozen afterwards drops out of the plan (it must project for itself to get its gradient)
    monkeypatch.setattr(ptp_utils, "CTX_KV_BATCHED", True)
    victim = plan["mods"][0]
    victim.to_k.weight.requires_grad_(True)
    try:
        with torch.no_grad():
            ptp_utils.find_pred_noise(ldm, both, cg, device=dev, noise=noise.cuda(), early_exit=True, controllers=controllers)
            controller.reset()
        assert victim not in ldm.unet._skp_ctx_plan["mods"] and len(ldm.unet._skp_ctx_plan["mods"]) == len(plan["mods"]) - 1
    finally:
        victim.to_k.weight.requires_grad_(False)


@pytest.mark.parametrize("N,C,H,W", [(8, 1280, 16, 16), (2, 320, 64, 64), (8, 640, 32, 32)])
def test_one_pass_group_norm_statistics_far_from_zero(N, C, H, W):
    """One-pass GroupNorm (row in registers, two in-register passes for mean and variance) on rows whose mean dwarfs their spread
    (mean 50, spread 0.1: a one-sweep sum / sum-of-squares form loses every digit of the variance here), forward and backward
    against fp64; repeat runs bit-identical."""
    from stablekeypoints_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, C, H, W, generator=g) * 0.1 + 50.0
    w = torch.randn(N, C, H, W, generator=g)
    norm = torch.nn.GroupNorm(32, C, eps=1e-5)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g)); norm.bias.copy_(torch.randn(C, generator=g))
    xd = x.double().requires_grad_(True)
    nd = torch.nn.GroupNorm(32, C, eps=1e-5).double()
    nd.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    ref = torch.nn.functional.silu(nd(xd))
    (ref * w.double()).sum().backward()
    norm = norm.cuda()
    outs = []
    for _ in range(2):
        xg = x.cuda().requires_grad_(True)
        y = ops.group_norm_silu(xg, norm)
        (y * w.cuda()).sum().backward()
        outs.append((y.detach().clone(), xg.grad.clone()))
    # the input itself is only known to 50 * 2^-24 = 3e-6 against a spread of 0.1: 3e-5 relative in xhat
    torch.testing.assert_close(outs[0][0].cpu().double(), ref.detach(), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(outs[0][1].cpu().double(), xd.grad, rtol=1e-3, atol=2e-4 * xd.grad.abs().max().item())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("N,C,H,W,silu,with_off", [(2, 320, 64, 64, True, False), (2, 1280, 16, 16, False, True), (1, 1920, 32, 32, True, True),
                                                   (2, 128, 128, 128, True, False)])
def test_group_norm_fork_adds_the_residual_gradient_in_kernel(N, C, H, W, silu, with_off):
    """(y, x') = fork(x): y = GroupNorm(+SiLU)(x), x' = x for the block's residual path; the gradient of x is the norm's input
    gradient PLUS the residual path's, added inside the norm's backward kernel (one-pass and two-kernel forms) -- against
    plain autograd over the un-forked composition, bit for bit in the forward, to rounding in the gradient."""
    from stablekeypoints_amd import ops
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(N, C, H, W, generator=g).cuda()
    off = torch.randn(N, C, generator=g).cuda() if with_off else None
    norm = torch.nn.GroupNorm(32, C).cuda()
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g)); norm.bias.copy_(torch.randn(C, generator=g))
    for p_ in norm.parameters():
        p_.requires_grad = False
    w1, w2 = torch.randn(N, C, H, W, generator=g).cuda(), torch.randn(N, C, H, W, generator=g).cuda()
    xa = x0.clone().requires_grad_(True)
    y, xp = ops.group_norm_silu_fork(xa, norm, off=off, silu=silu)
    assert type(y.grad_fn).__name__.startswith("GroupNormSiLUForkFn")
    ((y * w1).sum() + (xp * xp * w2).sum()).backward()
    xb = x0.clone().requires_grad_(True)
    y2 = ops.group_norm_silu(xb, norm, off=off, silu=silu)
    ((y2 * w1).sum() + (xb * xb * w2).sum()).backward()
    assert torch.equal(y, y2) and torch.equal(xp, xa)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-6 * xb.grad.abs().max().item())
    # only one of the two outputs used
    xc = x0.clone().requires_grad_(True)
    y3, xp3 = ops.group_norm_silu_fork(xc, norm, off=off, silu=silu)
    (xp3 * w2).sum().backward()
    torch.testing.assert_close(xc.grad, w2)
    xd = x0.clone().requires_grad_(True)
    y4, _ = ops.group_norm_silu_fork(xd, norm, off=off, silu=silu)
    (y4 * w1).sum().backward()
    xe = x0.clone().requires_grad_(True)
    (ops.group_norm_silu(xe, norm, off=off, silu=silu) * w1).sum().backward()
    assert torch.equal(xd.grad, xe.grad)


@pytest.mark.parametrize("strategy", ["gaussian", "entropy", "consistent"])
def test_batched_token_statistics_and_selection_equal_the_per_image_calls(strategy, monkeypatch):
    """Token statistics of all images in one launch, of all affine copies in another, the selection of all images in a third
    (skp_select_tokens_batched): the same selections, loss sums and q / k gradients as the per-image launches, bit for bit
    (ops.MapLossesFn on fixed hooked-layer inputs: every kernel on this path is deterministic)."""
    from oracle import ref_path as R
    from stablekeypoints_amd import ops
    from stablekeypoints_amd.optimize import token_order
    g = torch.Generator().manual_seed(21)
    n, T, H, Rr = 3, 21, 8, 128
    dims = [(16, 1280), (16, 1280), (32, 640)]
    thetas = [R.affine_matrix(7.0 * (i - 1), 0.9, (0.1 * i, -0.05)).reshape(-1).tolist() for i in range(n)]
    qs0 = [torch.randn(2 * n, s_ * s_, c, generator=g).cuda() for s_, c in dims]
    ks0 = [torch.randn(1, T, c, generator=g).cuda() for _, c in dims]
    meta = dict(R=Rr, heads=H, scales=[(c // H) ** -0.5 for _, c in dims], thetas=thetas, sigma=2.0, num_subjects=1,
                strategy=strategy, n_cand=12, top_k=5, score_fn=token_order)
    got = []
    for mode in (False, True):
        monkeypatch.setattr(ops, "MAP_LOSSES_BATCHED", mode)
        qs = [q.clone().requires_grad_(True) for q in qs0]
        ks = [k.clone().requires_grad_(True) for k in ks0]
        flat = []
        for q, k in zip(qs, ks):
            flat += [q, k]
        sh, eq, sel = ops.MapLossesFn.apply(meta, *flat)
        (3.0 * sh + 7.0 * eq).backward()
        got.append([sh.detach().clone(), eq.detach().clone(), sel.clone()] + [t.grad.clone() for t in qs + ks])
    for a_, b_ in zip(got[0], got[1]):
        assert torch.equal(a_, b_)
    assert got[0][2].shape == (n, 5) and len(set(got[0][2][0].tolist())) == 5
