#!/usr/bin/env python
"""Split-bf16 flash attention forward (csrc/skp_flash_attn_s.hip) against the fp32-instruction forward: time (events,
interleaved rounds, the split time INCLUDES its K / V pre-pass) and error vs fp64 on a slice."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablekeypoints_amd import ops

def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

rows = []
for name, B, H, N, d in (("64^2 layer, 8 rows, d=40", 8, 8, 4096, 40), ("64^2 layer, 2 rows, d=40", 2, 8, 4096, 40),
                         ("32^2 layer, 8 rows, d=80", 8, 8, 1024, 80), ("sd21 96^2/2 layer, 8 rows, d=80?", 8, 10, 2304, 80)):
    g = torch.Generator().manual_seed(0)
    C = H * d
    q, k, v = (torch.randn(B, N, C, generator=g).cuda() for _ in range(3))
    scale = d ** -0.5
    out32 = torch.empty_like(q); lse32 = torch.empty(B, H, N, device="cuda")
    f32 = lambda: ops.N.check(ops.N.lib().skp_flash_attn_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out32.data_ptr(), lse32.data_ptr(),
                                                                  B, B, H, N, N, d, float(scale), ops._stream()), "f32")
    spl = lambda: ops.flash_attn_fwd_split(q, k, v, H, scale)
    for _ in range(3): f32(); spl()
    t32, tsp = [], []
    for _ in range(5):
        t32.append(timeit(f32, 10)); tsp.append(timeit(spl, 10))
    t32.sort(); tsp.sort()
    # error on one (row, head): fp64 reference
    qd = q[0, :, :d].double(); kd = k[0, :, :d].double(); vd = v[0, :, :d].double()
    ref = ((qd @ kd.T) * scale).softmax(-1) @ vd
    f32(); o_s, _ = spl()
    e32 = (out32[0, :, :d].double() - ref).abs().max().item(); esp = (o_s[0, :, :d].double() - ref).abs().max().item()
    fl = 4.0 * N * N * d * B * H / 1e6
    row = dict(shape=name, B=B, H=H, N=N, d=d, f32_us=t32[2], split_us=tsp[2], speedup=t32[2] / tsp[2], f32_tf=fl / t32[2], split_tf_equiv=fl / tsp[2],
               f32_err=e32, split_err=esp, err_ratio=esp / e32)
    rows.append(row)
    print(f"{name:36s} f32 {t32[2]:8.1f} us {row['f32_tf']:6.1f} TF/s | split {tsp[2]:8.1f} us {row['split_tf_equiv']:6.1f} TF/s-eq | x{row['speedup']:.2f} | err {e32:.2e} / {esp:.2e} ({row['err_ratio']:.2f}x)", flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)

# ---- backward at d = 40 (the 64^2 layers): split (images + dQ kernel + dK/dV kernel) vs the fp32-instruction backward ----
for name, B, H, N, d in (("bwd 64^2 layer, 8 rows, d=40", 8, 8, 4096, 40), ("bwd 64^2 layer, 2 rows, d=40", 2, 8, 4096, 40)):
    g = torch.Generator().manual_seed(1)
    C = H * d
    q, k, v, go = (torch.randn(B, N, C, generator=g).cuda() for _ in range(4))
    scale = d ** -0.5
    out, lse = ops.flash_attn_fwd_split(q, k, v, H, scale)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    nb = ops.N.lib().skp_flash_attn_bwd_workspace(B, B, H, N, N, d)
    ws = torch.empty(nb // 4, device="cuda")
    f32 = lambda: ops.N.check(ops.N.lib().skp_flash_attn_bwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), go.data_ptr(), lse.data_ptr(),
                                                                  dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), B, B, H, N, N, d, float(scale),
                                                                  ops._stream()), "f32 bwd")
    spl = lambda: ops.flash_attn_bwd_split(q, k, v, out, go, lse, H, scale)
    for _ in range(3): f32(); spl()
    t32, tsp = [], []
    for _ in range(5):
        t32.append(timeit(f32, 5)); tsp.append(timeit(spl, 5))
    t32.sort(); tsp.sort()
    fl = 10.0 * N * N * d * B * H / 1e6                         # 2.5 x the forward's FLOPs
    row = dict(shape=name, B=B, H=H, N=N, d=d, f32_us=t32[2], split_us=tsp[2], speedup=t32[2] / tsp[2], f32_tf=fl / t32[2], split_tf_equiv=fl / tsp[2])
    rows.append(row)
    print(f"{name:36s} f32 {t32[2]:8.1f} us {row['f32_tf']:6.1f} TF/s | split {tsp[2]:8.1f} us {row['split_tf_equiv']:6.1f} TF/s-eq | x{row['speedup']:.2f}", flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
