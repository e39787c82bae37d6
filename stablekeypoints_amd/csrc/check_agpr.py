#!/usr/bin/env python3
"""Build-time guard for the NAMED accumulators of skp_conv_wino4.hip (skp_wino4_common.h: positions 0-31 of the 36 Winograd
accumulators live in a[0:255] through inline-assembly MFMAs; the compiler is told the AGPR file is clobbered by every such
statement, but nothing in the language stops a future compiler from parking a value of its own there BETWEEN two statements).

usage: check_agpr.py <object.o>      (run by the Makefile after skp_conv_wino4.o is built; non-zero exit fails the build)

For every kernel of the object that uses AGPRs it checks, on the disassembly of the gfx950 code object:
  * the only instructions that touch an AGPR are v_mfma_* (accumulate in place), v_accvgpr_read_b32 (epilogue) and
    v_accvgpr_write_b32 aN, 0 (the zero fill) -- i.e. no compiler-generated VGPR<->AGPR copy, no v_accvgpr_mov, no load / store /
    LDS instruction with an AGPR operand;
  * every MFMA with an AGPR destination accumulates into the same tuple it reads (a[n:n+3] ... a[n:n+3]);
  * no scratch: private_segment_fixed_size == 0 and vgpr_spill_count == 0.
The fp64 comparison tests (tests/test_kernels_gpu.py, test_round3_gpu.py) stay mandatory for every toolchain bump; this guard
turns the silent failure mode into a build error."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def main(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        run("objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat)
        targets = run(os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={fat}").split()
        dev = [t for t in targets if "amdgcn" in t]
        if not dev:
            sys.exit(f"check_agpr: no device code object in {obj}")
        run(os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={dev[0]}", f"--output={co}")
        asm = run(os.path.join(LLVM, "llvm-objdump"), "-d", co)
        notes = run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
    meta, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(agpr_count|name|private_segment_fixed_size|vgpr_spill_count):\s*(\S+)", line)
        if not m:
            continue
        if m.group(1) == "agpr_count" and cur:
            meta[cur.get("name")] = cur
            cur = {}
        cur[m.group(1)] = m.group(2)
    if cur:
        meta[cur.get("name")] = cur
    errors, checked, kernel = [], 0, None
    agpr = re.compile(r"\ba(\d+|\[\d+:\d+\])")
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel = m.group(1)
            continue
        if kernel is None or int(meta.get(kernel, {}).get("agpr_count", "0")) == 0:
            continue
        ins = line.split("//")[0].strip()
        if not ins or not agpr.search(ins):
            continue
        op, _, rest = ins.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        if op.startswith("v_mfma_"):
            if ops[0].startswith("a") and ops[0] != ops[3].split()[0]:
                errors.append(f"{kernel}: MFMA does not accumulate in place: {ins}")
        elif op == "v_accvgpr_read_b32":
            pass
        elif op == "v_accvgpr_write_b32":
            if ops[1] != "0":
                errors.append(f"{kernel}: AGPR written from a register (compiler-generated copy?): {ins}")
        else:
            errors.append(f"{kernel}: unexpected instruction on an AGPR: {ins}")
    for name, m in meta.items():
        if int(m.get("agpr_count", "0")) == 0:
            continue
        checked += 1
        if int(m.get("private_segment_fixed_size", "0")) or int(m.get("vgpr_spill_count", "0")):
            errors.append(f"{name}: scratch in a named-accumulator kernel (private {m.get('private_segment_fixed_size')}, spills {m.get('vgpr_spill_count')})")
    if not checked:
        errors.append("no AGPR kernel found (did the kernels move?)")
    if errors:
        print("check_agpr: FAILED\n  " + "\n  ".join(errors[:20]), file=sys.stderr)
        sys.exit(1)
    print(f"check_agpr: {checked} named-accumulator kernels clean (AGPRs touched only by MFMA / accvgpr_read / zero fill, no scratch)")


if __name__ == "__main__":
    main(sys.argv[1])
