#!/bin/bash
# Round 6 run D: is the systematic atomic-vs-deterministic offset of the fp32x3 golden gradients a regression?  the probe on the round-5 tree
# (build/r05_tree, a git worktree of f7fea92 with its own library) and on the current tree with single changes switched off.
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/r06_d_x3_probe.txt
: > $o
export PROBE_CASES=subset PROBE_MODES=fp32x3
echo "== round-5 tree (f7fea92)" >> $o; (cd build/r05_tree && python tools/probes/x3_atomic_vs_det_probe.py 2>&1 | grep -v amdgpu | cut -c1-900 >> $o)
echo "== current tree" >> $o; python tools/probes/x3_atomic_vs_det_probe.py 2>&1 | grep -v amdgpu | cut -c1-900 >> $o
echo "== current tree, DOSE_HIP_PASS_ARENA=0" >> $o; DOSE_HIP_PASS_ARENA=0 python tools/probes/x3_atomic_vs_det_probe.py 2>&1 | grep -v amdgpu | cut -c1-900 >> $o
echo "== current tree, DP_NO_TICKET=1" >> $o; DP_NO_TICKET=1 python tools/probes/x3_atomic_vs_det_probe.py 2>&1 | grep -v amdgpu | cut -c1-900 >> $o
echo "== current tree, ctypes + python apply" >> $o; DOSE_HIP_CTYPES=1 DOSE_HIP_PY_APPLY=1 python tools/probes/x3_atomic_vs_det_probe.py 2>&1 | grep -v amdgpu | cut -c1-900 >> $o
echo "== current tree, DP_HK_BUF=0" >> $o; DP_HK_BUF=0 python tools/probes/x3_atomic_vs_det_probe.py 2>&1 | grep -v amdgpu | cut -c1-900 >> $o
echo "== current tree, no side streams" >> $o; DOSE_HIP_WGRAD_STREAM=0 DOSE_HIP_BRANCH_STREAM=0 python tools/probes/x3_atomic_vs_det_probe.py 2>&1 | grep -v amdgpu | cut -c1-900 >> $o
cat $o
bash tools/r06_graph_repro.sh
