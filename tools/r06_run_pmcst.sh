#!/bin/bash
# instruction counters of the forward kernels with the round-6 slab staging against -DDP_CC16_BUF=0 -DDP_TILED_TAB=0 (build/ab): one PMC pass each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_pmcst
OLD=$PWD/build/ab/libdose_hip_stagingold.so
for v in new old; do
  if [ $v = old ]; then export DOSE_HIP_LIB=$OLD; else unset DOSE_HIP_LIB; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d ${o}_$v -o p --output-format csv -- python3 tools/bench_conv.py fwd > ${o}_$v.log 2>&1
  python tools/pmc_stalls.py ${o}_$v k_conv > ${o}_counters_$v.md 2>&1
done
unset DOSE_HIP_LIB
rm -rf ${o}_new ${o}_old
head -30 ${o}_counters_new.md | cut -c1-220; head -30 ${o}_counters_old.md | cut -c1-220; tail -3 ${o}_new.log
