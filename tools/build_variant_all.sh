#!/bin/bash
# usage: tools/build_variant_all.sh <name> <extra hipcc flags...> -> mquant_amd/libmquant_hip_<name>.so with EVERY translation unit rebuilt with the flags
set -e
cd "$(dirname "$0")/../mquant_amd/csrc"
NAME=$1; shift
mkdir -p /tmp/variant_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable"
pids=()
for f in *.hip; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f -o /tmp/variant_$NAME/${f%.hip}.o &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmquant_hip_$NAME.so /tmp/variant_$NAME/*.o
ls -la ../libmquant_hip_$NAME.so
