#!/bin/bash
# k_gemm_tn_grouped with three blocks per CU (__launch_bounds__(256, 3): 172 -> 164 VGPRs) against two: kernel time under the tracer, then the step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_g3
OLD=$PWD/build/ab/libdose_hip_gemmold.so
timeout 600 rocprofv3 --kernel-trace --stats -d ${o}_trace_new -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-fp32-leg --no-cpu-baseline > ${o}_trace_new.log 2>&1
python tools/prof_summary.py ${o}_trace_new 2>/dev/null | grep -i "gemm_tn\|gemm_nt" > ${o}_gemm_new.txt
export DOSE_HIP_LIB=$OLD
timeout 600 rocprofv3 --kernel-trace --stats -d ${o}_trace_old -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-fp32-leg --no-cpu-baseline > ${o}_trace_old.log 2>&1
unset DOSE_HIP_LIB
python tools/prof_summary.py ${o}_trace_old 2>/dev/null | grep -i "gemm_tn\|gemm_nt" > ${o}_gemm_old.txt
echo new; cat ${o}_gemm_new.txt; echo old; cat ${o}_gemm_old.txt
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -m gpu -x -q -k "linear or Linear or gemm or vit or ViT or g7 or g4 or mlp or deferred or grouped" 2>&1 | grep -E "passed|failed" | tail -3
for i in 1 2 3; do
  python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_new_$i.json
  DOSE_HIP_LIB=$OLD python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_old_$i.json
done
python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_new.json
DOSE_HIP_LIB=$OLD python bench.py --dtype fp32x3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_x3_old.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_g3_bench_line_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f"%r["ms_per_step"])
PY
