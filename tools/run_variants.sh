#!/bin/bash
# run a command once per experiment library in build_exp/ (MRIRT_LIB selects the shared object)
# usage: bash tools/run_variants.sh "<variant names>" <command...>
VARS=$1; shift
for v in $VARS; do
  echo "=== $v"
  MRIRT_LIB=$GRAFT_REPO_ROOT/build_exp/libmrirt_$v.so "$@" 2>&1 | grep -v amdgpu.ids
done
