#!/usr/bin/env python3
"""Per-kernel wave-cycle breakdown from one rocprofv3 --pmc pass of
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
(WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; MI355X_MICROARCH.md, PMC slots).
usage: tools/pmc_stalls.py <pmc_dir> [name filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    if flt and flt not in k:
        continue
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        cnt[k] += 1
names = sorted({c for v in per.values() for c in v})
print("| kernel | launches | " + " | ".join(n.replace("SQ_", "") for n in names) + " |")
print("|---|---|" + "---|" * len(names))
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1.0
    cells = []
    for n in names:
        x = v.get(n, 0)
        cells.append(f"{x / cnt[k]:.3g} ({100 * x / wc:.0f}%)")
    print(f"| {k} | {cnt[k]} | " + " | ".join(cells) + " |")
