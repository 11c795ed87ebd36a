#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_e
for i in 1 2 3; do echo "== poison probe, fp32x3 atomic, process $i" >> ${o}_poison.txt; python tools/probes/poison_probe.py fp32x3 4 multi 2>&1 | grep -v amdgpu | cut -c1-1500 >> ${o}_poison.txt; done
echo "== poison probe, fp32x3 deterministic" >> ${o}_poison.txt; DOSE_HIP_DETERMINISTIC=1 python tools/probes/poison_probe.py fp32x3 3 multi 2>&1 | grep -v amdgpu | cut -c1-1500 >> ${o}_poison.txt
echo "== poison probe, fp32 atomic" >> ${o}_poison.txt; python tools/probes/poison_probe.py fp32 3 multi 2>&1 | grep -v amdgpu | cut -c1-1500 >> ${o}_poison.txt
echo "== poison probe, bf16 atomic" >> ${o}_poison.txt; python tools/probes/poison_probe.py bf16 3 multi 2>&1 | grep -v amdgpu | cut -c1-1500 >> ${o}_poison.txt
cat ${o}_poison.txt
DOSE_HIP_CAPTURE_BRANCH=1 PYTHONFAULTHANDLER=1 timeout 300 python -X faulthandler bench.py --graph --steps 3 --warmup 2 --no-fp32-leg --no-cpu-baseline > ${o}_graph_branch.log 2>&1; echo "EXIT $?" >> ${o}_graph_branch.log
grep -v "^$" ${o}_graph_branch.log | tail -40 | cut -c1-250
DP_HK_BUF=1 python tools/bench_conv.py wgrad --filter "dec" 2>&1 | grep -v amdgpu > ${o}_wgrad_hk_frontloaded.txt; cat ${o}_wgrad_hk_frontloaded.txt
python tools/host_profile.py 2>&1 | head -9 > ${o}_host_profile.txt; cat ${o}_host_profile.txt
python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line.json; python tools/show_bench.py ${o}_bench_line.json | head -3
