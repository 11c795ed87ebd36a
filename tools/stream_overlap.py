#!/usr/bin/env python3
"""Per-queue busy time and overlap of the last full step in a rocprofv3 --kernel-trace csv directory (step marker: k_patchify
forward).  Answers: how much of the side stream's (ViT branch) kernel time runs beside main-stream kernels?
usage: tools/stream_overlap.py <trace_dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
marks = [e[0] for e in ev if "k_patchify" in e[2] and ("false" in e[2] or "chunks" in e[2])]
s0, s1 = marks[-2], marks[-1]
sel = [e for e in ev if s0 <= e[0] < s1]


def union(iv):
    iv = sorted(iv)
    out, cs, ce = [], None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s > ce:
            out.append((cs, ce)); cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        out.append((cs, ce))
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


byq = collections.defaultdict(list)
for s, e, n, q in sel:
    byq[q].append((s, e))
print(f"step span {1e-6 * (s1 - s0):.2f} ms, {len(sel)} kernels, union busy {1e-6 * length(union([(s, e) for s, e, _, _ in sel])):.2f} ms")
qs = sorted(byq, key=lambda q: -length(union(byq[q])))
us = {q: union(byq[q]) for q in qs}
for q in qs:
    print(f"queue {q}: {len(byq[q])} kernels, busy {1e-6 * length(us[q]):.2f} ms, sum of durations {1e-6 * sum(e - s for s, e in byq[q]):.2f} ms")
if len(qs) >= 2:
    main, side = qs[0], qs[1]
    ov = intersect(us[main], us[side])
    print(f"side-queue time beside main-queue kernels: {1e-6 * ov:.2f} ms of {1e-6 * length(us[side]):.2f} ms")
    # side-queue kernels: duration when overlapped vs alone
    c = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for s, e, n, q in sel:
        if q != side:
            continue
        k = n.replace("(anonymous namespace)::", "").split("(")[0][:50]
        o = intersect([(s, e)], us[main])
        if o > 0.5 * (e - s):
            c[k][0] += 1; c[k][1] += e - s
        else:
            c[k][2] += 1; c[k][3] += e - s
    print("side-queue kernel                                   overlapped: n, avg us | alone: n, avg us")
    for k, (n1, t1, n2, t2) in sorted(c.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:12]:
        print(f"  {k:50s} {n1:4d} {1e-3 * t1 / max(n1, 1):7.1f} | {n2:4d} {1e-3 * t2 / max(n2, 1):7.1f}")

if len(qs) >= 2 and len(sys.argv) > 2:
    # timeline: main-queue idle gaps longer than 20 us, with the neighbours and what the side queue did meanwhile
    mainev = sorted((s, e, n) for s, e, n, q in sel if q == main)
    sideev = sorted((s, e, n) for s, e, n, q in sel if q == side)
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    for (s_a, e_a, n_a), (s_b, e_b, n_b) in zip(mainev, mainev[1:]):
        if s_b - e_a > 20000:
            inside = [short(n) for s, e, n in sideev if s < s_b and e > e_a]
            print(f"t={1e-6 * (e_a - s0):7.2f} ms  main idle {1e-3 * (s_b - e_a):7.1f} us  after {short(n_a)}  before {short(n_b)}  side: {len(inside)} kernels {collections.Counter(inside).most_common(3)}")
