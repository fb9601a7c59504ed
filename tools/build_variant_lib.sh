#!/bin/bash
# usage: [FILE=gemm_ws] tools/build_variant_lib.sh <name> <extra hipcc flags...> -> mquant_amd/libmquant_hip_<name>.so: $FILE.hip rebuilt
# with the flags, every other object from the regular build.  Select with MQUANT_HIP_LIB.  (How the A/B files under profiles/
# were produced: experiment switches or patched working copies of one translation unit.)
set -e
cd "$(dirname "$0")/../mquant_amd/csrc"
FILE=${FILE:-gemm_ws}
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $FILE.hip -o /tmp/${FILE}_$NAME.o
OBJS=$(ls *.o | grep -v "^$FILE.o\$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmquant_hip_$NAME.so $OBJS /tmp/${FILE}_$NAME.o
ls -la ../libmquant_hip_$NAME.so
