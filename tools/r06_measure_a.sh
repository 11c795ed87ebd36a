#!/bin/bash
# Round 6, measurement pass A (no code change): host profile with / without the reducer, wave-cycle stall counters of k_wgrad_hk<7>, and the
# LDS phase counters of the 128^3-level 3x3x3 kernels with the DP_DBG knock-outs.  outputs: gpurun_out/r06_a_*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_a
python tools/host_profile.py > ${o}_host_profile.txt 2>&1
python tools/host_profile.py ddp > ${o}_host_profile_ddp.txt 2>&1
# k_wgrad_hk<7>: where the non-MFMA 45 % goes
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
C2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
C3="SQ_WAVE_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
for dbg in 0 1 2; do
  for ci in 1 2 3; do
    eval "C=\$C$ci"
    DP_DBG=$dbg timeout 300 rocprofv3 --pmc $C -d ${o}_wg_d${dbg}_c${ci} -o p --output-format csv -- python3 tools/bench_conv.py wgrad --filter "dec" > ${o}_wg.log 2>&1
    echo "## k_wgrad DP_DBG=$dbg counter set $ci" >> ${o}_wgrad_hk_stalls.md
    python tools/pmc_stalls.py ${o}_wg_d${dbg}_c${ci} k_wgrad >> ${o}_wgrad_hk_stalls.md 2>&1
  done
done
DP_DBG=0 python tools/bench_conv.py wgrad --filter "dec" > ${o}_wgrad_times.txt 2>&1
DP_DBG=1 python tools/bench_conv.py wgrad --filter "dec" >> ${o}_wgrad_times.txt 2>&1
DP_DBG=2 python tools/bench_conv.py wgrad --filter "dec" >> ${o}_wgrad_times.txt 2>&1
# 3x3x3 at the 128^3 level: LDS counters per phase
for dbg in 0 1 2 4 3 6 5 7; do
  for ci in 1 3; do
    eval "C=\$C$ci"
    DP_DBG=$dbg timeout 300 rocprofv3 --pmc $C -d ${o}_c3_d${dbg}_c${ci} -o p --output-format csv -- python3 tools/bench_conv.py fwd --filter "@128" > ${o}_c3.log 2>&1
    echo "## 128^3-level forward kernels, DP_DBG=$dbg counter set $ci" >> ${o}_3x3x3_phase_counters.md
    python tools/pmc_stalls.py ${o}_c3_d${dbg}_c${ci} k_conv >> ${o}_3x3x3_phase_counters.md 2>&1
  done
  echo "DP_DBG=$dbg" >> ${o}_c3_times.txt
  DP_DBG=$dbg python tools/bench_conv.py fwd --filter "@128" >> ${o}_c3_times.txt 2>&1
done
find gpurun_out -name '*.csv' -size +5M -delete 2>/dev/null
rm -rf ${o}_wg_d* ${o}_c3_d*
ls -la gpurun_out | grep r06_a
