#!/usr/bin/env python3
"""Per-stream timeline of the last bench step in a rocprofv3 --kernel-trace csv directory: busy time per HIP stream, and the idle
gaps of the busiest ("main") stream with what ran elsewhere during each gap -- i.e. where the critical path waits for a side stream.
usage: tools/stream_timeline.py <trace dir> [min gap us]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in rows)
marks = [e[0] for e in ev if "k_patchify" in e[2] and ("false" in e[2] or "chunks" in e[2])]
marks = [m for i, m in enumerate(marks) if i == 0 or m - marks[i - 1] > 5_000_000]
s0, s1 = marks[-2], marks[-1]
sel = [e for e in ev if s0 <= e[0] < s1]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


by = collections.defaultdict(list)
for e in sel:
    by[e[3]].append(e)
print(f"step span {1e-6 * (s1 - s0):.2f} ms, {len(sel)} kernels, {len(by)} streams")
busy = {}
for s, lst in by.items():
    busy[s] = sum(e[1] - e[0] for e in lst)
    print(f"  stream {s}: {len(lst):4d} kernels, busy {1e-6 * busy[s]:6.2f} ms, from {1e-6 * (lst[0][0] - s0):6.2f} to {1e-6 * (lst[-1][1] - s0):6.2f} ms")
main = max(busy, key=busy.get)
lst = by[main]
print(f"idle gaps of stream {main} longer than {min_gap:.0f} us (what the other streams ran meanwhile):")
tot = 0
for a, b in zip(lst, lst[1:]):
    gap = b[0] - a[1]
    if gap > 1e3 * min_gap:
        tot += gap
        others = collections.Counter()
        for e in sel:
            if e[3] != main and e[1] > a[1] and e[0] < b[0]:
                others[(e[3], short(e[2]))] += min(e[1], b[0]) - max(e[0], a[1])
        desc = ", ".join(f"[{s}] {k} {1e-3 * v:.0f}us" for (s, k), v in others.most_common(3))
        print(f"  at {1e-6 * (a[1] - s0):6.2f} ms: {1e-3 * gap:6.0f} us  after {short(a[2])[:34]:34s} before {short(b[2])[:34]:34s} | {desc}")
print(f"total {1e-6 * tot:.2f} ms in such gaps")
# time during which nothing at all runs
allb = sorted((e[0], e[1]) for e in sel)
idle, ce = 0, allb[0][1]
for s, e in allb[1:]:
    if s > ce:
        idle += s - ce
    ce = max(ce, e)
print(f"whole-GPU idle inside the step: {1e-6 * idle:.2f} ms")
