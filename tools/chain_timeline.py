#!/usr/bin/env python3
"""The transformer's backward chain in a rocprofv3 --kernel-trace csv directory (last bench step): every kernel of the stream that runs the
attention backward kernels, from the first k_attn_bwd-side kernel of the step to the stream's last kernel -- start (ms after the step's
start), duration, gap to the previous kernel on that stream, and what share of the chain window is kernels / gaps.
usage: tools/chain_timeline.py <trace dir> [list]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in rows)
marks = [e[0] for e in ev if "k_patchify" in e[2] and ("false" in e[2] or "chunks" in e[2])]
marks = [m for i, m in enumerate(marks) if i == 0 or m - marks[i - 1] > 5_000_000]
s0, s1 = marks[-2], marks[-1]
sel = [e for e in ev if s0 <= e[0] < s1]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


bwd = [e for e in sel if "k_attn_bwd" in e[2] or "k_attn_delta" in e[2]]
if not bwd:
    sys.exit("no attention backward kernel in the step")
stream = bwd[0][3]
t_first = bwd[0][0]
lst = [e for e in sel if e[3] == stream]
# the chain starts with the stream's first kernel after its forward part: the last kernel before a pause of > 2 ms ends the forward
start_i = 0
for i in range(1, len(lst)):
    if lst[i][0] <= t_first and lst[i][0] - lst[i - 1][1] > 2_000_000:
        start_i = i
chain = lst[start_i:]
span = chain[-1][1] - chain[0][0]
busy = sum(e[1] - e[0] for e in chain)
gaps = [b[0] - a[1] for a, b in zip(chain, chain[1:])]
print(f"stream {stream}: chain of {len(chain)} kernels from {1e-6 * (chain[0][0] - s0):.2f} to {1e-6 * (chain[-1][1] - s0):.2f} ms of a {1e-6 * (s1 - s0):.2f} ms step: "
      f"window {1e-6 * span:.2f} ms = kernels {1e-6 * busy:.2f} + gaps {1e-6 * (span - busy):.2f} (median gap {1e-3 * sorted(gaps)[len(gaps) // 2]:.1f} us)")
agg = collections.defaultdict(lambda: [0, 0])
for e in chain:
    a = agg[short(e[2])]
    a[0] += 1
    a[1] += e[1] - e[0]
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {1e-3 * t:8.1f} us  {n:4d} x {1e-3 * t / n:7.1f} us  {k}")
# what the other streams ran during the chain window
oth = collections.Counter()
for e in sel:
    if e[3] != stream and e[1] > chain[0][0] and e[0] < chain[-1][1]:
        oth[e[3]] += min(e[1], chain[-1][1]) - max(e[0], chain[0][0])
print("other streams busy inside the window:", ", ".join(f"[{s}] {1e-6 * v:.2f} ms" for s, v in oth.most_common()))
if len(sys.argv) > 2:
    prev = None
    for e in chain:
        print(f"{1e-6 * (e[0] - s0):8.3f} ms  {1e-3 * (e[1] - e[0]):7.1f} us  gap {1e-3 * (e[0] - prev) if prev else 0:6.1f} us  {short(e[2])}")
        prev = e[1]
