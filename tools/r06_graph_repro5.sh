#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_graph_repro5.txt
: > $o
for pat in 97 99; do c=0; t=0; for n in 6 10 20 40; do for seed in $(seq 1 50); do
  out=$(timeout 20 tools/probes/graph_fork_join_repro.bin $pat $n 0 $seed 2>&1); rc=$?; t=$((t+1))
  if [ $rc -ne 0 ]; then c=$((c+1)); echo "pattern $pat n=$n seed=$seed EXIT $rc :: $(echo $out | head -c 300)" >> $o; fi
done; done; echo "pattern $pat: $c of $t random graphs crash" >> $o; done
grep "random graphs crash" $o; grep "pattern 97" $o | head -5
