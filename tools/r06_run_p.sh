#!/bin/bash
cd "$GRAFT_REPO_ROOT"
DOSE_DDP_TIMING=1 python tools/host_profile.py ddp > gpurun_out/r06_p_host_profile_ddp.txt 2>&1
grep -n "====" -A70 gpurun_out/r06_p_host_profile_ddp.txt | cut -c1-200 | head -90
