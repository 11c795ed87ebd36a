#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export DOSE_HIP_CAPTURE_BRANCH=1
timeout 600 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop" -ex run -ex "bt 40" -ex "info sharedlibrary hip" --args python3 tools/probes/graph_capture_bisect.py fwdbwd 64 > gpurun_out/r06_l_gdb.txt 2>&1
grep -v "^\[New Thread\|^\[Thread\|amdgpu.ids" gpurun_out/r06_l_gdb.txt | tail -70 | cut -c1-260
