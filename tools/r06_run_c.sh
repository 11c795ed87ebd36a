#!/bin/bash
# Round 6 run C: buffer-load staging of k_wgrad_hk<7> (A/B + parity), fast C binding (A/B on the host profile), bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_c
for v in 0 1; do
  echo "DP_HK_BUF=$v" >> ${o}_wgrad_hk_ab.txt
  DP_HK_BUF=$v python tools/bench_conv.py wgrad --filter "dec" 2>&1 | grep -v amdgpu >> ${o}_wgrad_hk_ab.txt
done
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py::test_conv_sampled_oracle_full_size tests/test_round6_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "conv or wgrad or grad or round6 or dice or nested or folded or deterministic" 2>&1 | grep -v "^$" | tail -30 > ${o}_tests.txt
python tools/probes/x3_atomic_vs_det_probe.py > ${o}_x3_atomic_vs_det.txt 2>&1
python tools/host_profile.py > ${o}_host_profile.txt 2>&1
DOSE_HIP_CTYPES=1 DOSE_HIP_PY_APPLY=1 python tools/host_profile.py 2>&1 | head -12 > ${o}_host_profile_ctypes.txt
python tools/host_profile.py ddp 2>&1 | head -14 > ${o}_host_profile_ddp.txt
python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line.json
DP_HK_BUF=0 python bench.py --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > ${o}_bench_line_hkbuf0.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d ${o}_mfma -o p --output-format csv -- python3 tools/bench_conv.py wgrad --filter "dec" > ${o}_mfma.log 2>&1
python tools/pmc_mfma.py ${o}_mfma ${o}_wgrad_mfma_busy.md ${o}_wgrad_mfma_busy.json > /dev/null 2>&1
rm -rf ${o}_mfma
cat ${o}_x3_atomic_vs_det.txt; cat ${o}_wgrad_hk_ab.txt; tail -5 ${o}_tests.txt; head -8 ${o}_host_profile.txt; head -8 ${o}_host_profile_ctypes.txt
python tools/show_bench.py ${o}_bench_line.json 2>/dev/null | head -5
