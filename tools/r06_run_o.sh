#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_o
for v in 0 1; do echo "DP_HK_BUF=$v" >> ${o}_wgrad3_ab.txt; DP_HK_BUF=$v python tools/bench_conv.py wgrad --filter "c3" 2>&1 | grep -v amdgpu >> ${o}_wgrad3_ab.txt; done
cat ${o}_wgrad3_ab.txt
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py::test_conv_sampled_oracle_full_size tests/test_x3_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "conv or wgrad or grad" 2>&1 | grep -v "^$" | tail -6 > ${o}_tests.txt; cat ${o}_tests.txt
python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line.json; python tools/show_bench.py ${o}_bench_line.json | head -8
( export DOSE_DDP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611; python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_rccl_one_rank.json ); python tools/show_bench.py ${o}_bench_line_rccl_one_rank.json | head -1
python - <<'PY'
import json
for f in ("gpurun_out/r06_o_bench_line.json","gpurun_out/r06_o_bench_line_rccl_one_rank.json"):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "host", [round(x,1) for x in r["host_enqueue_ms_per_step"]], r["config"].get("backward_on_calling_thread"), r["config"].get("c_binding"))
PY
