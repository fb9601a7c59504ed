#!/bin/bash
# usage: tools/build_variant_lib.sh <name> <extra hipcc flags...> -> mquant_amd/libmquant_hip_<name>.so: gemm_ws.hip rebuilt with
# the flags, every other object from the regular build.  Select with MQUANT_HIP_LIB.  (How the A/B files under profiles/ were
# produced while experiment switches existed in the source; today it serves patched working copies of gemm_ws.hip.)
set -e
cd "$(dirname "$0")/../mquant_amd/csrc"
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable"
/opt/rocm/bin/hipcc $FLAGS "$@" -c gemm_ws.hip -o /tmp/gemm_ws_$NAME.o
OBJS=$(ls *.o | grep -v '^gemm_ws.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmquant_hip_$NAME.so $OBJS /tmp/gemm_ws_$NAME.o
ls -la ../libmquant_hip_$NAME.so
