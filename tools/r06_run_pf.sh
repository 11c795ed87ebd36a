#!/bin/bash
# token GEMM (k_gemm_nt<2,2,128>): K steps in flight in registers, 3 (built) / 2 / 1 (round 4)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_pf
for v in pf3 pf2 pf1; do
  if [ $v = pf3 ]; then unset DOSE_HIP_LIB; else export DOSE_HIP_LIB=$PWD/build/ab/libdose_hip_gemm$v.so; fi
  NO_REF=1 python tools/bench_gemm.py 2>&1 | grep " us " > ${o}_bench_gemm_$v.txt
done
unset DOSE_HIP_LIB
paste -d'|' ${o}_bench_gemm_pf3.txt ${o}_bench_gemm_pf2.txt ${o}_bench_gemm_pf1.txt | sed 's/(torch.matmul[^|]*//g' | cut -c1-200
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -m gpu -x -q -k "linear or Linear or gemm or vit or ViT or g7 or g4 or mlp or attention" 2>&1 | grep -E "passed|failed" | tail -3
for i in 1 2; do
  for v in pf3 pf2 pf1; do
    if [ $v = pf3 ]; then unset DOSE_HIP_LIB; else export DOSE_HIP_LIB=$PWD/build/ab/libdose_hip_gemm$v.so; fi
    python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_${v}_$i.json
  done
done
unset DOSE_HIP_LIB
python bench.py --model transeg --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_transeg_pf3.json
DOSE_HIP_LIB=$PWD/build/ab/libdose_hip_gemmpf1.so python bench.py --model transeg --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_transeg_pf1.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_pf_bench_line_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f"%r["ms_per_step"])
PY
