#!/bin/bash
# The evidence set of one code state (run through gpurun): bench lines of every configuration, in-step / serial / fp32x3 / OAR-TRANSEG / C5
# per-step kernel tables, MFMA-busy and FETCH / WRITE PMC passes of the default step, deterministic-mode and one-rank-RCCL bench lines,
# the row-length micro-benchmark.   usage: tools/evidence_round.sh <tag>      outputs: gpurun_out/<tag>_*
tag=$1
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.txt 2>&1; tail -2 gpurun_out/${tag}_smoke.txt
tools/bench_all_configs.sh ${tag} > gpurun_out/${tag}_all.txt 2>&1
DOSE_HIP_DETERMINISTIC=1 python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/${tag}_bench_line_deterministic.json
( export DOSE_DDP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611; python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/${tag}_bench_line_rccl_one_rank.json )
tools/profile_round.sh ${tag} all > /dev/null 2>&1
tools/profile_round.sh ${tag}_serial trace --no-side-stream > /dev/null 2>&1
tools/profile_round.sh ${tag}_x3serial trace --dtype fp32x3 --no-side-stream > /dev/null 2>&1
tools/profile_round.sh ${tag}_transeg_serial trace --model transeg --no-side-stream > /dev/null 2>&1
tools/profile_round.sh ${tag}_c5_serial trace --model cascade --dtype fp16 --size 192 192 128 --batch 1 --roi 96 --checkpoint --loss-scale 1024 --no-side-stream > /dev/null 2>&1
python tools/bench_conv.py all --widths > gpurun_out/${tag}_bench_conv_widths.txt 2>&1
python tools/bench_conv.py all > gpurun_out/${tag}_bench_conv.txt 2>&1
( export DOSE_DDP_ALGO=rs_ag DOSE_DDP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612; python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/${tag}_bench_line_rccl_one_rank_rs_ag.json )
python bench.py --engine-thread --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/${tag}_bench_line_engine_thread.json
python tools/host_profile.py > gpurun_out/${tag}_host_profile.txt 2>&1
python tools/host_profile.py ddp > gpurun_out/${tag}_host_profile_ddp.txt 2>&1
rm -rf gpurun_out/${tag}*_trace gpurun_out/${tag}_mfma gpurun_out/${tag}_fetch gpurun_out/${tag}_write
cat gpurun_out/${tag}_all.txt | grep -v amdgpu
ls gpurun_out | grep ${tag} | head -60
