#!/usr/bin/env python3
"""Per-step kernel-time table from a rocprofv3 --kernel-trace csv directory of bench.py (step marker: k_patchify forward)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marks = [e[0] for e in ev if "k_patchify" in e[2] and ("false" in e[2] or "chunks" in e[2])]
# (the fp32x3 mode patchifies three operand blocks back to back: keep the first mark of each burst)
marks = [m for i, m in enumerate(marks) if i == 0 or m - marks[i - 1] > 5_000_000]
s0, s1 = marks[-2], marks[-1]
sel = [e for e in ev if s0 <= e[0] < s1]
busy = 0
cs, ce = sel[0][0], sel[0][1]
for s, e, _ in sel[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"step span {1e-6 * (s1 - s0):.2f} ms, GPU busy {1e-6 * busy:.2f} ms, {len(sel)} kernels")
c, n = collections.Counter(), collections.Counter()
for s, e, nm in sel:
    k = nm.replace("(anonymous namespace)::", "").split("(")[0][:70]
    c[k] += e - s
    n[k] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for k, v in c.most_common(top):
    print(f"{v / 1e6:7.2f} ms {n[k]:5d}  {k}")
