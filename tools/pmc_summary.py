#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately: the TCC block cannot
hold both).  Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): both counters are in
KiB of memory-side requests; FETCH_SIZE tallies 128-B requests as 64 B for 16-byte-per-lane streaming reads, so it is DOUBLED.
usage: tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [<out.md>]"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return per


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fe) | set(wr)):
        f, w = fe.get(k, []), wr.get(k, [])
        n = max(len(f), len(w))
        fb = 2.0 * 1024 * sum(v for v, _ in f) / max(1, len(f))          # bytes per launch (gfx950: x2)
        wb = 1024.0 * sum(v for v, _ in w) / max(1, len(w))
        ns = sum(t for _, t in f) / max(1, len(f))
        rows[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                   "avg_ns_under_pmc": ns}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 2 --warmup 1",
               "correction": "FETCH_SIZE x2 (gfx950), KiB -> bytes", "kernels": rows}, open(out, "w"), indent=1)
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as md:
            md.write("| kernel | launches | fetch MB/launch (x2 corrected) | write MB/launch | GB/s under PMC |\n|---|---|---|---|---|\n")
            for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:40]:
                bw = v["hbm_bytes_per_launch"] / max(1.0, v["avg_ns_under_pmc"])
                md.write(f"| `{k[:80]}` | {v['launches']} | {v['fetch_bytes_per_launch'] / 1e6:.1f} | {v['write_bytes_per_launch'] / 1e6:.1f} | {bw:.0f} |\n")


if __name__ == "__main__":
    main()
