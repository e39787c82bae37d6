#!/bin/bash
# Counter passes (never combined with a trace) over the persistent Winograd kernel, ONE launch shape per process so that the
# per-kernel averages of tools/pmc_summary.py are per shape:  tools/pmc_conv.sh <tag> [shapes...]
#   -> gpurun_out/pmc_conv_<tag>/<shape>.json  (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_MFMA;
#      FETCH_SIZE; WRITE_SIZE -- three passes per shape) and summary.md
# shapes = names of tools/ab_build.py (default: gn s256 s128 u320 = the three VAE levels in their in-step forms + 320->320 @64^2)
set -u
TAG=$1; shift
SHAPES=${@:-gn s256 s128 u320}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_conv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in $SHAPES; do
  for p in 1 2 3; do
    case $p in
      1) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA";;
      2) C="FETCH_SIZE";;
      3) C="WRITE_SIZE";;
    esac
    rocprofv3 --pmc $C -f csv -d $OUT/$s/p$p -o p -- python $ROOT/tools/ab_build.py --worker $s 4 > $OUT/$s.p$p.log 2>&1
  done
  python $ROOT/tools/pmc_summary.py $OUT/$s.json $(find $OUT/$s -name "*counter_collection.csv") > $OUT/$s.txt 2>&1
done
cd $ROOT
python - "$OUT" $SHAPES <<'PY'
import json, sys, os
out, shapes = sys.argv[1], sys.argv[2:]
SH = {"gn": (8, 128, 128, 512), "plain": (8, 128, 128, 512), "s256": (8, 256, 256, 256), "s128": (8, 512, 512, 128), "s64": (8, 512, 512, 64),
      "u320": (8, 320, 320, 64), "u1280_16": (8, 1280, 1280, 16), "u640_32": (8, 640, 640, 32), "u1280_8": (8, 1280, 1280, 8),
      "r1280_8": (8, 1280, 1280, 8), "r2560_8": (8, 2560, 1280, 8), "r1280_16": (8, 1280, 1280, 16), "r2560_16": (8, 2560, 1280, 16)}
lines = ["| shape | kernel | MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)) | HBM-side bytes / launch (2 x FETCH + WRITE) | algorithmic bytes | ratio |", "|---|---|---|---|---|---|"]
for s in shapes:
    try:
        k = json.load(open(os.path.join(out, s + ".json")))["kernels"]
    except Exception as e:
        lines.append(f"| {s} | (no data: {e}) | | | | |"); continue
    B, ci, co, sz = SH[s]
    raw = s.startswith("r")                      # raw-filter form: 9 taps instead of 36 transformed values (+ its helper kernels' rows)
    alg = 4 * (B * ci * sz * sz + B * co * sz * sz + (9 if raw else 36) * ci * co)
    for name, c in k.items():
        if "skp_wino4_conv" not in name and "skp_wino4r_" not in name and not (raw and "skp_wino4_reduce" in name):
            continue
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, 4 * c.get("SQ_BUSY_CU_CYCLES", 0)) if "SQ_BUSY_CU_CYCLES" in c else float("nan")
        tr = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        lines.append(f"| {s}: {ci}->{co} @{sz}^2, {B} rows | {name} | {busy:.3f} | {tr / 1e6:.0f} MB | {alg / 1e6:.0f} MB | {tr / alg:.2f} |")
open(os.path.join(out, "summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*.csv" -size +2M -delete
