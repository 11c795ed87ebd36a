#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the conv micro-benchmark (tools/bench_conv.py) for one filter: tools/pmc_conv.sh <tag> <filter> [env...]
tag=$1; filt=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/$tag
timeout 300 rocprofv3 --pmc FETCH_SIZE -d ${o}_fetch -o p --output-format csv -- python3 tools/bench_conv.py fwd --filter "$filt" > ${o}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d ${o}_write -o p --output-format csv -- python3 tools/bench_conv.py fwd --filter "$filt" > ${o}_write.log 2>&1
python tools/pmc_summary.py ${o}_fetch ${o}_write ${o}_pmc.json ${o}_pmc.md
grep "k_conv" ${o}_pmc.md
find ${o}_fetch ${o}_write -name '*.csv' -size +5M -delete 2>/dev/null
