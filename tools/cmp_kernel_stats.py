"""Per-kernel time difference between two `rocprofv3 --kernel-trace --stats -f csv` runs of the same command (A/B of a switch).
   python tools/cmp_kernel_stats.py <dir A> <dir B> <steps the runs timed>"""
import csv, sys, glob, re
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
def cls(n):
    if n.startswith("Cijk"): return "gemm"
    if "fa2" in n or "fas_" in n: return "flash"
    if "wino" in n or "conv" in n: return "conv"
    if "elementwise" in n or "copy" in n.lower(): return "elementwise/copy"
    return "other"
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3])
rows = []
tot = {}
for n in set(a) | set(b):
    ca, ta = a.get(n, (0, 0.0)); cb, tb = b.get(n, (0, 0.0))
    rows.append(((tb - ta) / steps / 1e6, n, ca, cb, ta / steps / 1e6, tb / steps / 1e6))
    c = tot.setdefault(cls(n), [0.0, 0.0, 0, 0]); c[0] += ta / steps / 1e6; c[1] += tb / steps / 1e6; c[2] += ca; c[3] += cb
rows.sort()
print("delta ms/step (B - A) | calls A, B | ms/step A, B | kernel")
for r in rows[:12] + rows[-12:]:
    print(f"{r[0]:+8.3f} | {r[2]:5d} {r[3]:5d} | {r[4]:8.3f} {r[5]:8.3f} | {r[1][:110]}")
for k, v in sorted(tot.items()):
    print(f"class {k:18s} A {v[0]:8.3f} ms/step ({v[2]} calls)  B {v[1]:8.3f} ms/step ({v[3]} calls)  delta {v[1] - v[0]:+.3f}")
