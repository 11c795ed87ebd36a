#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/probes/x3_split_sites_probe.py 2>&1 | grep -v amdgpu | cut -c1-330 > gpurun_out/r06_s_x3_split_sites.txt; cat gpurun_out/r06_s_x3_split_sites.txt | head -70
