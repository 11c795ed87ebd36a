#!/bin/bash
# Round 6 run B: measured spread of the atomic-mode golden gates, the new / changed tests, a bench line with the new objects.
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_b
python tools/golden_spread.py 200 > ${o}_golden_spread.txt 2>&1
cp gpurun_out/x3_atomic_spread.json tests/golden/x3_atomic_spread.json
timeout 1500 python -m pytest tests/test_round6_gpu.py tests/test_parity128_gpu.py::test_pyfer_128_fp32x3_backward_matches_oracle \
  "tests/test_models_gpu.py" "tests/test_x3_gpu.py::test_reference_goldens_in_x3_mode" \
  "tests/test_ops_gpu.py::test_conv_epilogue_statistics" -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v "^$" | tail -120 > ${o}_tests.txt
python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line.json
( export DOSE_DDP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611; python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_rccl_one_rank.json )
( export DOSE_DDP_ALGO=rs_ag DOSE_DDP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612; python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_rccl_one_rank_rs_ag.json )
tail -30 ${o}_tests.txt
