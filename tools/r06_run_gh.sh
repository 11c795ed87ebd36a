#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/r06_graph_repro2.sh
bash tools/r06_run_g.sh
