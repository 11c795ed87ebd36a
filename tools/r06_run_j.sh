#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_j_x3_backward_diff.txt
: > $o
for i in 1 2 3; do echo "== process $i" >> $o; PROBE_FROM=34 PROBE_TO=64 python tools/probes/x3_backward_diff_probe.py 2>&1 | grep -v amdgpu | cut -c1-400 >> $o; done
cat $o
