#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_u
timeout 1500 python -m pytest tests/test_round6_gpu.py tests/test_parity128_gpu.py tests/test_models_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | grep -v "^$" | grep "passed\|failed\|rror\|assert" | tail -8 > ${o}_tests.txt; cat ${o}_tests.txt
python tools/probes/x3_split_sites_probe.py 2>&1 | grep -v amdgpu | cut -c1-200 | head -4
for i in 1 2; do python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_$i.json; done
python bench.py --model cascade --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_cascade.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_u_bench_line_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f"%r["ms_per_step"])
PY
