#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_graph_repro3.txt
: > $o
for n in 3 4 6 10 20; do for seed in $(seq 1 40); do
  out=$(timeout 20 tools/probes/graph_fork_join_repro.bin 99 $n 0 $seed 2>&1); rc=$?
  if [ $rc -ne 0 ]; then echo "n=$n seed=$seed EXIT $rc :: $out" | head -c 600 >> $o; echo >> $o; fi
done; done
echo "crashing configurations: $(grep -c EXIT $o) of 200" >> $o
head -30 $o; tail -2 $o
