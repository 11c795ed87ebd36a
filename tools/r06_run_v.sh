#!/bin/bash
# k_conv_cc16 staging through buffer loads: parity first, then the per-shape bench and the step, new vs -DDP_CC16_BUF=0 (build/ab)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_v
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py tests/test_round6_gpu.py tests/test_round5_gpu.py tests/test_parity128_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -5 > ${o}_tests.txt
cat ${o}_tests.txt
OLD=$PWD/build/ab/libdose_hip_cc16old.so
python tools/bench_conv.py fwd > ${o}_bench_conv_all_new.txt 2>&1
DOSE_HIP_LIB=$OLD python tools/bench_conv.py fwd > ${o}_bench_conv_all_old.txt 2>&1
for i in 1 2; do
  python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_new_$i.json
  DOSE_HIP_LIB=$OLD python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_old_$i.json
done
python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_new.json
DOSE_HIP_LIB=$OLD python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_old.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_v_bench_line_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f"%r["ms_per_step"])
PY
paste -d'|' gpurun_out/r06_v_bench_conv_all_new.txt gpurun_out/r06_v_bench_conv_all_old.txt | cut -c1-260 | tail -40
