#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_y
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -12
python tools/probes/backward_tail_probe.py --no-adam-overlap > ${o}_tail_stepalone.txt 2>&1
python tools/probes/backward_tail_probe.py > ${o}_tail_overlap.txt 2>&1
tail -12 ${o}_tail_stepalone.txt; tail -12 ${o}_tail_overlap.txt
DOSE_HIP_VIT_EARLY_GROUP=0 python tools/probes/backward_tail_probe.py > ${o}_tail_overlap_nogroups.txt 2>&1; tail -10 ${o}_tail_overlap_nogroups.txt
for g in 1 2 4; do
  DOSE_HIP_VIT_EARLY_GROUP=$g python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_overlap_g$g.json
done
DOSE_HIP_VIT_EARLY_GROUP=0 python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_overlap_g0.json
python bench.py --no-adam-overlap --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_stepalone_g2.json
DOSE_HIP_VIT_EARLY_GROUP=0 python bench.py --no-adam-overlap --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_stepalone_g0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_y_bench_line_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f"%r["ms_per_step"], "early/step", r["config"].get("adam_early_updates_per_step"), "host", [round(x,1) for x in r.get("host_enqueue_ms_per_step",[])][:3])
PY
