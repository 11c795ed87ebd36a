#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_q
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "allreduce or rccl or bench_two or vit_backward or streams or stream" 2>&1 | grep -v "^$" | tail -5 > ${o}_tests.txt; cat ${o}_tests.txt
python tools/host_profile.py ddp 2>&1 | head -12 > ${o}_host_profile_ddp.txt; cat ${o}_host_profile_ddp.txt | grep -v amdgpu
python tools/host_profile.py 2>&1 | grep "host enqueue" > ${o}_host_profile.txt; cat ${o}_host_profile.txt
( export DOSE_DDP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611; python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_rccl_one_rank.json )
python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line.json
python - <<'PY'
import json
for f in ("gpurun_out/r06_q_bench_line.json","gpurun_out/r06_q_bench_line_rccl_one_rank.json"):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/step %.2f"%r["ms_per_step"], "host", [round(x,1) for x in r["host_enqueue_ms_per_step"]])
PY
