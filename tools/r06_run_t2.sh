#!/bin/bash
# k_conv_tiled staging from a per-block voxel table + buffer loads: parity, then steps and per-shape bench new vs -DDP_TILED_TAB=0 (build/ab)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_t2
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py tests/test_round6_gpu.py tests/test_round5_gpu.py tests/test_parity128_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > ${o}_tests.txt
cat ${o}_tests.txt
OLD=$PWD/build/ab/libdose_hip_tiledold.so
python tools/bench_conv.py fwd --dtype fp32x3 > ${o}_bench_conv_x3_new.txt 2>&1
DOSE_HIP_LIB=$OLD python tools/bench_conv.py fwd --dtype fp32x3 > ${o}_bench_conv_x3_old.txt 2>&1
python tools/bench_conv.py fwd > ${o}_bench_conv_bf16_new.txt 2>&1
DOSE_HIP_LIB=$OLD python tools/bench_conv.py fwd > ${o}_bench_conv_bf16_old.txt 2>&1
for i in 1 2; do
  python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_new_$i.json
  DOSE_HIP_LIB=$OLD python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_old_$i.json
done
python bench.py --dtype fp16 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_f16_new.json
DOSE_HIP_LIB=$OLD python bench.py --dtype fp16 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_f16_old.json
python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_new.json
DOSE_HIP_LIB=$OLD python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_old.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_t2_bench_line_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f"%r["ms_per_step"])
PY
paste -d'|' ${o}_bench_conv_x3_new.txt ${o}_bench_conv_x3_old.txt | cut -c1-260 | tail -24
paste -d'|' ${o}_bench_conv_bf16_new.txt ${o}_bench_conv_bf16_old.txt | cut -c1-260 | tail -34
