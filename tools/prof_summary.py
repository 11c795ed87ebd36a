#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats (csv) output directory into a short per-kernel table (markdown)."""
import csv
import glob
import sys


def main(d, out=None, top=25):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        name = r["Name"].replace("|", "/")
        if len(name) > 90:
            name = name[:87] + "..."
        lines.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
