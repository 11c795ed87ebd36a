#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_f_graph_bisect.txt
: > $o
run() { echo "== $*" >> $o; ( env "$@" timeout 200 python tools/probes/graph_capture_bisect.py $PH 64 2>&1 | grep -v amdgpu | grep "CAPTURE OK\|REPLAY OK\|Error\|error\|Fatal\|capture_end\|line" | head -8 | cut -c1-200 >> $o ); echo "   exit ${PIPESTATUS[0]}" >> $o; }
for PH in fwd_nograd fwd fwdbwd step; do
  run DOSE_HIP_CAPTURE_BRANCH=1
done
PH=fwdbwd
run DOSE_HIP_CAPTURE_BRANCH=1 DOSE_HIP_WGRAD_STREAM=0
run DOSE_HIP_CAPTURE_BRANCH=1 DOSE_HIP_SIDE_STREAM=0
run DOSE_HIP_CAPTURE_BRANCH=1 DOSE_HIP_WGRAD_STREAM=0 DOSE_HIP_SIDE_STREAM=0
run DOSE_HIP_CAPTURE_BRANCH=1 DOSE_HIP_PASS_ARENA=0
run DOSE_HIP_CAPTURE_BRANCH=0
PH=fwd
run DOSE_HIP_CAPTURE_BRANCH=1 DOSE_HIP_SIDE_STREAM=0
cat $o
