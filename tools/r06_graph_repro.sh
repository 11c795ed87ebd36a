#!/bin/bash
# matrix of the stand-alone fork / join capture repro (each configuration in its own process: a segfault must not end the sweep)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_graph_repro.txt
: > $o
for mode in 0 2; do for reuse in 0 1; do for pat in 0 1 2 3 4; do for pairs in 8 30 100 400; do
  timeout 30 tools/probes/graph_fork_join_repro.bin $pat $pairs $mode $reuse >> $o 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "pattern $pat pairs $pairs mode $mode reuse $reuse: EXIT $rc" >> $o; fi
done; done; done; done
# the real thing: the training step captured with the branch stream allowed (DOSE_HIP_CAPTURE_BRANCH=1), in a child process
DOSE_HIP_CAPTURE_BRANCH=1 timeout 300 python bench.py --graph --steps 5 --warmup 3 --no-fp32-leg --no-cpu-baseline > gpurun_out/r06_graph_branch.log 2>&1; echo "bench --graph with the branch stream: EXIT $?" >> $o
timeout 300 python bench.py --graph --steps 10 --warmup 3 --no-fp32-leg --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06_graph_bench_line.json; echo "bench --graph (shipped): EXIT $?" >> $o
cat $o | tail -90
tail -5 gpurun_out/r06_graph_branch.log
