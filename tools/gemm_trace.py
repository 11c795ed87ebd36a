#!/usr/bin/env python3
"""Average k_gemm_nt dispatch duration per consecutive group of N launches in a rocprofv3 --kernel-trace of
`NO_REF=1 tools/bench_gemm.py` (true kernel time; the HIP-event loop in bench_gemm.py includes launch gaps)."""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 21
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"),
               r.get("Grid_Size_Z", "?")) for r in csv.DictReader(open(f)) if "k_gemm_nt" in r["Kernel_Name"])
for i in range(0, len(rows), n):
    g = rows[i:i + n][1:]
    if not g:
        continue
    avg = sum(e - s for s, e, *_ in g) / len(g) / 1e3
    print(f"group {i // n:2d}: {avg:7.1f} us  grid=({g[0][3]},{g[0][4]},{g[0][5]})  {g[0][2].split('(')[0][-40:]}")
