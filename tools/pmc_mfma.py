#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE [SQ_BUSY_CYCLES ...]).
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over every SIMD of the device (32 per v_mfma_f32_32x32x16_bf16,
16 per v_mfma_f32_16x16x32_bf16: MI355X_MICROARCH.md, cycle constants); GRBM_GUI_ACTIVE counts the cycles the dispatch kept the
GPU busy at the clock it actually ran at -- summed over the 8 XCDs (each has its own GRBM: the per-launch value is 8 x duration x
clock; checked against the dispatch timestamps, which give 1.9-2.1 GHz under MFMA load).
utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs).
usage: tools/pmc_mfma.py <pmc_dir> <out.md> [<out.json>]"""
import collections
import csv
import glob
import json
import sys

SIMDS = 256 * 4
XCDS = 8


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for k, c in per.items():
        mf, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("GRBM_GUI_ACTIVE", [])
        if not mf or not gui or sum(mf) == 0:
            continue
        n = len(mf)
        util = sum(mf) / (sum(gui) / XCDS * SIMDS)
        rows.append({"kernel": k, "launches": n, "mfma_busy_cycles_per_launch": sum(mf) / n, "gui_active_cycles_per_launch": sum(gui) / n,
                     "avg_us_under_pmc": sum(dur[k]) / max(1, len(dur[k])) / 1e3, "mfma_util": util,
                     "clock_ghz_under_pmc": (sum(gui) / n / XCDS) / max(1.0, sum(dur[k]) / max(1, len(dur[k])))})
    rows.sort(key=lambda r: -r["mfma_busy_cycles_per_launch"] * r["launches"])
    with open(out, "w") as md:
        md.write("| kernel | launches | MFMA-busy cycles / launch (all SIMDs) | GUI-active cycles / launch | MFMA utilisation | clock GHz | us / launch (under PMC) |\n"
                 "|---|---|---|---|---|---|---|\n")
        for r in rows[:60]:
            md.write(f"| `{r['kernel'][:90]}` | {r['launches']} | {r['mfma_busy_cycles_per_launch']:.3e} | {r['gui_active_cycles_per_launch']:.3e} | "
                     f"{100 * r['mfma_util']:.1f} % | {r['clock_ghz_under_pmc']:.2f} | {r['avg_us_under_pmc']:.1f} |\n")
    if len(sys.argv) > 3:
        json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of bench.py", "simds": SIMDS, "kernels": rows}, open(sys.argv[3], "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
