#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1200 python tools/probes/graph_repro_search.py 5 3 > gpurun_out/r06_graph_repro4.txt 2>&1
cat gpurun_out/r06_graph_repro4.txt | head -60
