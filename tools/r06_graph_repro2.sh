#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_graph_repro2.txt
: > $o
for pat in 5 6 7 8 9; do for pairs in 1 4 12 40; do for reuse in 0 1; do
  timeout 30 tools/probes/graph_fork_join_repro.bin $pat $pairs 0 $reuse >> $o 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "pattern $pat pairs $pairs mode 0 reuse $reuse: EXIT $rc" >> $o; fi
done; done; done
cat $o
