#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06_k_tests.txt
tail -8 gpurun_out/r06_k_tests.txt
