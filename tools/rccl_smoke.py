#!/usr/bin/env python
"""Single-rank RCCL smoke: the exchange step of the path (SUM all-reduce of the [1,T,768] embedding gradient, reference
optimize.py:405-406,422-425) issued through `backend="nccl"` (= RCCL on ROCm) with world_size 1 on the box's one MI355X.

What it proves without a second GPU: librccl loads and initialises a communicator on this box, the collective is ordered
against the HIP kernels of the step on the launch stream (the gradient is all-reduced WITHOUT a host synchronisation after
`group_step`; a copy taken on the same stream before the collective must equal the reduced buffer bit for bit), the
optimizer step that follows reads the reduced buffer, and what one call costs on an idle stream (us per call over `--calls`
back-to-back all-reduces, events on the launch stream).  Prints ONE JSON line."""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=1000)
    ap.add_argument("--tokens", type=int, default=77)
    a = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1)
    from stablekeypoints_amd import _native
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import default_args, group_step
    from stablekeypoints_amd.optimize_token import load_ldm
    _native.lib()
    ldm, controllers, _ = load_ldm("cuda:0", "tiny", feature_upsample_res=32)
    dev, controller = next(iter(controllers.items()))
    g = torch.Generator().manual_seed(0)
    n, T = 2, a.tokens
    images = torch.rand(n, 3, 128, 128, generator=g)
    ctx = torch.randn(1, T, 768, generator=g).cuda().requires_grad_(True)
    noise = torch.randn(2 * n, 4, 16, 16, generator=g).cuda()
    torch.manual_seed(3)                                         # the affine draws of group_step
    args = default_args(num_tokens=T, feature_upsample_res=32, furthest_point_num_samples=8, top_k=4, batch_size=n)
    opt = torch.optim.Adam([ctx], lr=5e-3)
    tr = RandomAffineWithInverse()
    first = dist.all_reduce(torch.ones(4, device="cuda"))        # communicator set-up happens on the first collective
    torch.cuda.synchronize()
    init_s = time.perf_counter() - t0
    ordered = True
    for it in range(5):                                          # kernels -> collective -> optimizer, no host sync in between
        group_step(ldm, images, ctx, args, controller, tr, denom=n, noise=noise)
        before = ctx.grad.clone()                                # same stream, before the collective
        dist.all_reduce(ctx.grad, op=dist.ReduceOp.SUM)
        after = ctx.grad.clone()
        prev = ctx.detach().clone()
        opt.step()
        opt.zero_grad(set_to_none=False)
        torch.cuda.synchronize()
        ordered &= bool(torch.equal(before, after)) and bool(before.abs().max() > 0) and not bool(torch.equal(prev, ctx.detach()))
    grad = torch.randn(1, T, 768, device="cuda")
    ref = grad.clone()
    for _ in range(20):
        dist.all_reduce(grad)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    e0.record()
    for _ in range(a.calls):
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - w0
    value_ok = bool(torch.equal(grad, ref))                      # SUM over one rank is the identity, bit for bit
    line = {"what": "single-rank RCCL all-reduce(SUM) of the embedding gradient on the step's stream",
            "backend": dist.get_backend(), "world_size": dist.get_world_size(), "elements": grad.numel(), "bytes": grad.numel() * 4,
            "calls": a.calls, "us_per_call_stream": e0.elapsed_time(e1) * 1e3 / a.calls, "us_per_call_host": wall * 1e6 / a.calls,
            "value_ok": value_ok, "ordered_against_step_kernels": ordered, "init_plus_first_collective_s": init_s,
            "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "device": torch.cuda.get_device_name(0)}
    print(json.dumps(line), flush=True)
    dist.destroy_process_group()
    if not (value_ok and ordered):
        sys.exit(1)


if __name__ == "__main__":
    main()
