#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06_c2
timeout 600 rocprofv3 --kernel-trace -d ${o}_trace -o p --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-fp32-leg --no-cpu-baseline > ${o}_trace.log 2>&1
python tools/chain_timeline.py ${o}_trace list > ${o}_chain_timeline.txt 2>&1
python tools/stream_timeline.py ${o}_trace 100 > ${o}_stream_timeline.txt 2>&1
timeout 600 rocprofv3 --kernel-trace -d ${o}_x3trace -o p --output-format csv -- python3 bench.py --dtype fp32x3 --steps 4 --warmup 2 --no-fp32-leg --no-cpu-baseline > ${o}_x3trace.log 2>&1
python tools/chain_timeline.py ${o}_x3trace > ${o}_x3_chain_timeline.txt 2>&1
rm -rf ${o}_trace ${o}_x3trace
head -40 ${o}_chain_timeline.txt; head -30 ${o}_x3_chain_timeline.txt; head -12 ${o}_stream_timeline.txt
