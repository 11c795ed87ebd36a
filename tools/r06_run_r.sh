#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/r06_graph_repro5.sh
o=gpurun_out/r06_r
for i in 1 2; do
  python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_new_$i.json
  DP_HK_BUF=0 DOSE_HIP_CTYPES=1 DOSE_HIP_PY_APPLY=1 python bench.py --engine-thread --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_r05like_$i.json
done
python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_new.json
DP_HK_BUF=0 python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_hkbuf0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_r_bench_line_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f"%r["ms_per_step"], "host", [round(x,1) for x in r["host_enqueue_ms_per_step"]])
PY
