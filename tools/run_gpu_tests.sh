#!/bin/bash
# Run every GPU test function in its own process (a GPU fault in one must not hide the others); log to gpurun_out/.
mkdir -p gpurun_out
out=gpurun_out/gpu_tests.txt
: > $out
for t in "$@"; do
  echo "=== $t" >> $out
  timeout 600 python -m pytest "$t" -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "^  File\|pluggy\|_pytest\|^$" | tail -${TAILN:-25} >> $out
done
cat $out
