#!/usr/bin/env python
"""Where do the library (ATen / MIOpen) kernels of one optimisation step come from?
Runs the bench step under torch.profiler (shapes + python stacks) and prints the non-skp device time grouped by
(operator, input shapes, innermost stablekeypoints_amd frame).
    python tools/aten_ops.py [--model sd15] [--top 40] [--match elementwise,copy,add,conv]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="sd15")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--skip", default="mm,addmm,bmm,baddbmm,linear,matmul", help="operators left out (library GEMMs)")
    a = ap.parse_args()
    from stablekeypoints_amd import tuning
    from stablekeypoints_amd.invertable_transform import RandomAffineWithInverse
    from stablekeypoints_amd.optimize import SyntheticImages, default_args, group_step
    from stablekeypoints_amd.optimize_token import load_ldm
    dev = torch.device("cuda", 0)
    ldm, controllers, _ = load_ldm(dev, a.model, feature_upsample_res=128, init_on_device=True)
    tuning.enable()
    size = {"sd21": 768, "sdxl": 1024}.get(a.model, 512)
    width = ldm.unet.config["cross_attention_dim"]
    args = default_args(num_tokens=77, feature_upsample_res=128, batch_size=a.images, device=str(dev), image_size=size)
    data = SyntheticImages(n=16, size=size, seed=0, device=dev)
    ctx = torch.randn(1, 77, width).to(dev).requires_grad_(True)
    opt = torch.optim.Adam([ctx], lr=args.lr)
    tr = RandomAffineWithInverse(args.augment_degrees, args.augment_scale, args.augment_translate)

    def step():
        images = torch.stack([data[i]["img"] for i in range(a.images)])
        group_step(ldm, images, ctx, args, controllers[dev], tr, denom=a.images)
        opt.step(); opt.zero_grad(set_to_none=True)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    skip = set(a.skip.split(","))
    groups = {}
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        t = sum(k.duration for k in ev.kernels)
        if not t or ev.name.split("::")[-1] in skip:
            continue
        kern = ",".join(sorted({k.name[:60] for k in ev.kernels}))
        if "skp_" in kern or "Cijk" in kern:
            continue
        frame = ""
        for fr in (ev.stack or []):
            if "stablekeypoints_amd" in fr or "bench.py" in fr or "tools/" in fr:
                frame = fr.split("stablekeypoints_amd/")[-1]
                break
        key = (ev.name, str(ev.input_shapes)[:90], frame[:70])
        g = groups.setdefault(key, [0.0, 0, kern])
        g[0] += t; g[1] += 1
    tot = sum(v[0] for v in groups.values())
    print(f"non-GEMM library device time in one step: {tot / 1e3:.2f} ms")
    for (name, shapes, frame), (t, n, kern) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print(f"{t / 1e3:7.3f} ms {n:4d}x  {name:<28} {shapes:<90} {frame}  [{kern[:50]}]")


if __name__ == "__main__":
    main()
