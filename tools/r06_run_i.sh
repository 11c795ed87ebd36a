#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_i_splitk.txt
: > $o
for i in 1 2 3; do echo "== process $i fp32x3" >> $o; python tools/probes/x3_splitk_probe.py fp32x3 2>&1 | grep -v amdgpu >> $o; done
echo "== fp32" >> $o; python tools/probes/x3_splitk_probe.py fp32 2>&1 | grep -v amdgpu >> $o
echo "== bf16" >> $o; python tools/probes/x3_splitk_probe.py bf16 2>&1 | grep -v amdgpu >> $o
cat $o
