#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_tail
python tools/probes/backward_tail_probe.py > ${o}_tail_default.txt 2>&1
tail -16 ${o}_tail_default.txt
