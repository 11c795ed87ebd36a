#!/bin/bash
# is the process-dependent offset of the fp32x3 atomic-mode gradients a missing stream dependency?  several processes per stream configuration
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_g_x3_streams.txt
: > $o
export PROBE_CASES=subset PROBE_MODES=fp32x3 PROBE_ATOMIC_ONLY=1
probe() { for i in 1 2 3 4; do ( env "$@" python tools/probes/x3_atomic_vs_det_probe.py 2>&1 | grep "atomic\|mask" | cut -c1-330 >> $o ); done; }
echo "== default streams" >> $o; probe A=1
echo "== every side stream off" >> $o; probe DOSE_HIP_SIDE_STREAM=0 DOSE_HIP_WGRAD_STREAM=0 DOSE_HIP_BRANCH_STREAM=0
echo "== transformer stream off only" >> $o; probe DOSE_HIP_SIDE_STREAM=0
echo "== stream probe off (first three pool streams)" >> $o; probe DOSE_HIP_STREAM_PROBE=0
unset PROBE_ATOMIC_ONLY
for m in 1 2 4 8 32; do echo "== default streams, deterministic site mask $m" >> $o; export PROBE_MASK=$m; probe A=1; done
cat $o
