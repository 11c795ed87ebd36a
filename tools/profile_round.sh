#!/bin/bash
# Full evidence set of one bench configuration on the GPU box (run through gpurun; program directly after `--`):
#   kernel trace + stats, MFMA-busy PMC pass, FETCH_SIZE and WRITE_SIZE PMC passes (each PMC list in its own run).
# usage: tools/profile_round.sh <tag> [trace|all] [bench.py args...]      outputs under gpurun_out/<tag>_*
tag=$1; what=${2:-all}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 4 --warmup 2 --no-fp32-leg --no-cpu-baseline $*"
o=gpurun_out/$tag
timeout 600 rocprofv3 --kernel-trace --stats -d ${o}_trace -o p --output-format csv -- $B > ${o}_trace.log 2>&1
python tools/step_profile.py ${o}_trace 100 > ${o}_per_step_kernel_time.txt
python tools/prof_summary.py ${o}_trace > ${o}_kernel_stats.md 2>/dev/null
if [ "$what" = all ]; then
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d ${o}_mfma -o p --output-format csv -- $B > ${o}_mfma.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d ${o}_fetch -o p --output-format csv -- $B > ${o}_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d ${o}_write -o p --output-format csv -- $B > ${o}_write.log 2>&1
  python tools/pmc_mfma.py ${o}_mfma ${o}_mfma_busy.md ${o}_mfma_busy.json > /dev/null
  python tools/pmc_summary.py ${o}_fetch ${o}_write ${o}_pmc_traffic.json ${o}_pmc_traffic.md
fi
# keep the summaries, drop the raw traces (the merge back is capped at 64 MiB)
find ${o}_trace ${o}_mfma ${o}_fetch ${o}_write -name '*.csv' -size +20M -delete 2>/dev/null
du -sh ${o}_* | tail -12
head -45 ${o}_per_step_kernel_time.txt
