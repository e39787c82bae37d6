#!/usr/bin/env python
"""Classify the kernels of a rocprofv3 `--kernel-trace -f csv` run of bench.py into kernel classes, over a
steady-state window: the last N optimisation steps, delimited by the Adam (multi_tensor_apply) launches.
usage: step_breakdown.py <run_kernel_trace.csv> [N=2] [conv_log.json]

With the launch list `bench.py --conv-log conv_log.json` wrote (shape + algorithmic FLOPs of every Winograd F(4x4,3x3) call of
one step, in launch order) the trace's `skp_wino4_conv*` launches are matched to it by position and the TIME-WEIGHTED fraction
of the fp32 matrix peak over ALL `skp_wino4_*` kernels of a step (K-split reductions included in the time) is printed:
sum of FLOPs / sum of kernel time / 157.3 TF/s -- the figure bench.py carries as roofline.conv_all_launches.frac."""
import csv
import json
import sys

F32_MATRIX_PEAK_TF = 157.3

CLASSES = [
    ("skp conv3x3 (Winograd stride 1 + direct stride 2)", ("skp_wino", "skp_conv_s2")),
    ("skp flash attention (self + long-key cross)", ("skp_self_attn", "skp_fa2_", "skp_fas_")),
    ("skp attention map fwd/bwd (north-star kernel)", ("skp_attn_map", "skp_map_")),
    ("skp fused GroupNorm+SiLU / bias+residual / add+LayerNorm", ("skp_group_norm", "skp_gn_", "skp_add_bias", "skp_add_ln")),
    ("skp cross-attn (T<=128) / selection / loss / small gemm / geglu / layout", ("skp_",)),
    ("conv (MIOpen)", ("igemm", "Igemm", "conv", "Conv", "winograd", "Winograd", "gridwise_convolution", "naive_conv",
                       "SubTensorOpWithScalar", "batched_transpose", "Im2Col", "Col2Im", "kernel_grouped_conv")),
    ("gemm (hipBLASLt/rocBLAS)", ("Cijk", "gemm", "Gemm")),
    ("softmax (ATen)", ("softmax", "SoftMax")),
    ("norm (ATen)", ("layer_norm", "LayerNorm", "GroupNorm", "RowwiseMoments")),
    ("elementwise / copy (ATen)", ("elementwise", "Elementwise", "vectorized", "copy", "Copy", "CatArray", "upsample",
                                   "index", "fill", "reduce", "Reduce")),
]


def main():
    path = sys.argv[1]
    nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    conv_log = json.load(open(sys.argv[3]))["launches"] if len(sys.argv) > 3 else None
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"],
                         row.get("Grid_Size") or "x".join(row.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))))
    rows.sort()
    # one optimizer step = one burst of multi_tensor_apply kernels; boundary = end of the last kernel of a burst
    bounds, last = [], None
    for st, en, name, _ in rows:
        if "multi_tensor_apply" in name:
            if last is not None and st - last < 2_000_000:
                bounds[-1] = en
            else:
                bounds.append(en)
            last = en
    t0, t1 = bounds[-1 - nwin], bounds[-1]
    tot, names, shapes = {}, {}, {}
    for st, en, name, grid in rows:
        if st < t0 or en > t1:
            continue
        if any(k in name for k in ("skp_wino", "skp_conv_s2", "skp_fa2_", "skp_self_attn", "skp_attn_map", "skp_map_", "Cijk")):
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:70]
            sh = shapes.setdefault((short, grid), [0.0, 0])
            sh[0] += en - st; sh[1] += 1
        for cls, keys in CLASSES:
            if any(k in name for k in keys):
                break
        else:
            cls = "other"
        tot[cls] = tot.get(cls, 0.0) + (en - st)
        d = names.setdefault(cls, {})
        a = d.setdefault(name[:100], [0.0, 0])
        a[0] += en - st; a[1] += 1
    total = sum(tot.values())
    print(f"window {(t1 - t0) / nwin / 1e6:.2f} ms/step, kernel time {total / nwin / 1e6:.2f} ms/step "
          f"(GPU busy {100 * total / (t1 - t0):.0f} %), {nwin} steady-state steps\n")
    print("| class | ms/step | % |\n|---|---|---|")
    for cls, ns in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"| {cls} | {ns / nwin / 1e6:.2f} | {100 * ns / total:.1f} |")
    print()
    for cls in ("conv (MIOpen)", "skp conv3x3 (Winograd stride 1 + direct stride 2)", "skp flash attention (self + long-key cross)", "elementwise / copy (ATen)",
                "other"):
        print(f"top kernels in '{cls}':")
        for n, (ns, c) in sorted(names.get(cls, {}).items(), key=lambda kv: -kv[1][0])[:8]:
            print(f"  {ns / nwin / 1e6:7.2f} ms/step  {c / nwin:6.1f} calls/step  {n}")
    print("\nper launch shape (kernel, grid threads): calls/step, average us, ms/step -- the rows `roofline.launch_us` of bench.py "
          "can be checked against")
    print("| kernel | grid | calls/step | avg us | ms/step |\n|---|---|---|---|---|")
    for (n, grid), (ns, c) in sorted(shapes.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"| {n} | {grid} | {c / nwin:.1f} | {ns / c / 1e3:.1f} | {ns / nwin / 1e6:.2f} |")
    # the 128-channel Winograd form is persistent (256 workgroups whatever the layer): its launches are told apart by their
    # position in the step's launch order, which is the same every step
    seqs = []
    for k in range(nwin):
        lo, hi = bounds[-1 - nwin + k], bounds[-nwin + k]
        seqs.append([(name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60], en - st)
                     for st, en, name, _ in rows if st >= lo and en <= hi and "skp_wino4_conv_c128_kernel" in name])
    if seqs and seqs[0] and all(len(q) == len(seqs[0]) for q in seqs):
        print("\npersistent 128-channel Winograd launches in launch order (# = position among this kernel's launches of a step; "
              "the VAE encoder comes first: #0-#3 are the 128 -> 128 convolutions at image resolution), 24 heaviest")
        print("| # | kernel form <STATS, GNF> | avg us |\n|---|---|---|")
        avg = [(i, seqs[0][i][0], sum(q[i][1] for q in seqs) / nwin / 1e3) for i in range(len(seqs[0]))]
        for i, n, us in sorted(sorted(avg, key=lambda r: -r[2])[:24]):
            print(f"| {i} | {n.replace('skp_wino4_conv_c128_kernel', '')} | {us:.1f} |")
    if conv_log:
        conv_fraction(rows, bounds, nwin, conv_log)


def conv_fraction(rows, bounds, nwin, conv_log):
    """Time-weighted fraction of the fp32 matrix peak over every skp_wino4_* kernel of a step."""
    short = lambda name: name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    tot_f = tot_ns = 0.0
    per = {}
    for k in range(nwin):
        lo, hi = bounds[-1 - nwin + k], bounds[-nwin + k]
        win = [(st, en, short(name)) for st, en, name, _ in rows if st >= lo and en <= hi and ("skp_wino4_" in name or "skp_wino4r_" in name)]
        is_main = lambda n: "skp_wino4_conv" in n or "skp_wino4r_conv" in n
        main = [w for w in win if is_main(w[2])]                     # one per C-ABI call; reduce / input-transform kernels only add time
        if len(main) != len(conv_log):
            print(f"\nconv log has {len(conv_log)} launches, step {k} of the trace {len(main)}: not the same configuration")
            return
        tot_ns += sum(en - st for st, en, _ in win)
        tot_f += sum(l["algorithmic_flops"] for l in conv_log)
        # attribute a reduce kernel to the conv launch in front of it, the raw-filter form's input transform to the one behind it
        cur, held = None, 0
        for st, en, name in win:
            if "skp_wino4r_input" in name:
                held += en - st
                continue
            if is_main(name):
                cur = conv_log[main.index((st, en, name))]
                key = (name.replace("skp_wino4_conv", "").replace("skp_wino4r_conv", "raw-filter"), cur["Cin"], cur["Cout"], cur["H"], cur["W"], cur["B"])
                d = per.setdefault(key, [0, 0.0, 0.0])
                d[0] += 1; d[1] += cur["algorithmic_flops"]
                d[2] += held
                held = 0
            if cur is not None:
                per[key][2] += en - st
    print(f"\nall skp_wino4_* / skp_wino4r_* kernels: {tot_f / nwin / 1e12:.3f} TFLOP executed (direct-form / 4) in {tot_ns / nwin / 1e6:.2f} ms per "
          f"step => {tot_f / tot_ns / 1e3:.1f} TF/s = {tot_f / tot_ns / 1e3 / F32_MATRIX_PEAK_TF:.3f} of the {F32_MATRIX_PEAK_TF} TF/s "
          "fp32 matrix peak (time-weighted over every launch)")
    print("| kernel form | Cin->Cout @ HxW, rows | calls/step | ms/step | frac |\n|---|---|---|---|---|")
    for (form, ci, co, H, W, B), (c, fl, ns) in sorted(per.items(), key=lambda kv: -kv[1][2])[:30]:
        print(f"| {form} | {ci}->{co} @ {H}x{W}, {B} | {c / nwin:.1f} | {ns / nwin / 1e6:.2f} | {fl / ns / 1e3 / F32_MATRIX_PEAK_TF:.3f} |")


if __name__ == "__main__":
    main()
