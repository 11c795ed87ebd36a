#!/bin/bash
# A/B library: the csrc files that differ from git revision $1 compiled from THAT revision and linked with the current objects of
# everything else -> build/ab/libdose_hip_ab.so (travels with gpurun; select it with DOSE_HIP_LIB=build/ab/libdose_hip_ab.so).
# usage: tools/build_ab.sh <rev>        (run python -c "import __graft_entry__ as g; g.build()" first)
set -e
rev=${1:-HEAD}
cd "$(dirname "$0")/.."
mkdir -p build/ab/src/dose_prediction_amd/csrc build/ab/src/include build/ab/obj
git show $rev:include/dose_hip.h > build/ab/src/include/dose_hip.h
git show $rev:dose_prediction_amd/csrc/common.h > build/ab/src/dose_prediction_amd/csrc/common.h
objs=""
for f in dose_prediction_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if git show $rev:$f > build/ab/src/$f 2>/dev/null && ! cmp -s $f build/ab/src/$f; then
    echo "[ab] $b from $rev"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -c build/ab/src/$f -o build/ab/obj/$b.o &
    objs="$objs build/ab/obj/$b.o"
  else
    objs="$objs build/obj/$b.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libdose_hip_ab.so $objs
ls -la build/ab/libdose_hip_ab.so
