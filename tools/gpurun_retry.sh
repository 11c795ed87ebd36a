#!/bin/bash
# gpurun with retries while no GPU slot is free (exit code 3 = nothing charged).  usage: tools/gpurun_retry.sh TIMEOUT 'command'
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  gpurun --timeout "$t" -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
