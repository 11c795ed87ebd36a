#!/usr/bin/env python3
"""Pretty-print the JSON line of a bench.py log (last line that parses)."""
import json
import sys

for line in reversed(open(sys.argv[1]).read().splitlines()):
    try:
        r = json.loads(line)
    except Exception:
        continue
    print(f"value={r['value']:.3f} {r['unit']}  ms/step={r['ms_per_step']:.1f}  gpus={r['n_gpus']} dtype={r['dtype']}")
    print("roofline:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r["roofline"].items()})
    for k, v in sorted(r.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
        print(f"  {k:22s} ms/step={v['ms_per_step']:8.2f}  launches={v['launches_per_step']:6.1f}  avg_ms={v['avg_launch_ms']:7.3f}  TF={v['tflops']:7.1f}")
    if "cpu_baseline" in r:
        print("cpu_baseline:", r["cpu_baseline"])
    break
else:
    print("no JSON line; tail:")
    print("\n".join(open(sys.argv[1]).read().splitlines()[-15:]))
