#!/bin/bash
# usage: tools/debug/attn_abl.sh "<variants>" -> tools/debug/attn_abl.py once per library variant (mquant_amd/libmquant_hip_<v>.so; main = the regular build)
cd "$GRAFT_REPO_ROOT"
for v in $1; do
  lib=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip_$v.so; [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip.so
  MQUANT_HIP_LIB=$lib python tools/debug/attn_abl.py 2>&1 | tail -1
done
