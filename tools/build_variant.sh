#!/bin/bash
# A/B library from the CURRENT sources with extra compiler flags on one file: tools/build_variant.sh <name> <file.hip> <flags...>
# -> build/ab/libdose_hip_<name>.so (select with DOSE_HIP_LIB=...; travels with gpurun)
set -e
name=$1; file=$2; shift; shift
cd "$(dirname "$0")/.."
mkdir -p build/ab/obj
b=$(basename $file .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC "$@" -c $file -o build/ab/obj/${b}_$name.o
objs=""
for f in dose_prediction_amd/csrc/*.hip; do bb=$(basename $f .hip); if [ $bb = $b ]; then objs="$objs build/ab/obj/${b}_$name.o"; else objs="$objs build/obj/$bb.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libdose_hip_$name.so $objs
ls -la build/ab/libdose_hip_$name.so
