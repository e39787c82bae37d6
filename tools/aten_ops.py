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
    ap.add_argument("--gemm", action="store_true", help="list the library GEMMs instead (operator, shapes, calls, us, TF/s)")
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
    if a.gemm:
        return gemm_table(prof, a.top)
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


def gemm_table(prof, top):
    """Library GEMM launches of the step grouped by (operator, operand shapes): calls, device us, fp32 TF/s."""
    groups = {}
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        if ev.name.split("::")[-1] not in ("mm", "addmm", "bmm", "baddbmm"):
            continue
        t = sum(k.duration for k in ev.kernels)
        shp = [s for s in ev.input_shapes if s]
        mats = [s for s in shp if len(s) >= 2][-2:]
        if len(mats) < 2:
            continue
        a_, b_ = mats
        batch = a_[0] if len(a_) == 3 else 1
        fl = 2.0 * batch * a_[-2] * a_[-1] * b_[-1]
        frame = ""
        for fr in (ev.stack or []):
            if "stablekeypoints_amd" in fr:
                frame = fr.split("stablekeypoints_amd/")[-1]
                break
        kern = ",".join(sorted({k.name[:44] for k in ev.kernels}))
        key = (ev.name.split("::")[-1], str(a_), str(b_), frame[:48])
        g = groups.setdefault(key, [0.0, 0, fl, kern])
        g[0] += t; g[1] += 1
    tot = sum(v[0] for v in groups.values())
    print(f"library GEMM device time in one step: {tot / 1e3:.2f} ms in {sum(v[1] for v in groups.values())} launches")
    for (name, sa, sb, frame), (t, n, fl, kern) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:top]:
        tf = fl * n / (t * 1e-6) / 1e12
        print(f"{t / 1e3:7.3f} ms {n:3d}x {t / n:7.1f} us {tf:6.1f} TF/s ({tf / 157.3:.2f}) {name:<6} {sa:<20} x {sb:<18} {frame:<48} [{kern}]")


if __name__ == "__main__":
    main()
