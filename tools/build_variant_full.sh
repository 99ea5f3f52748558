#!/bin/bash
# usage: tools/build_variant_full.sh <tag> [-DNAME=VALUE ...] -- lib/libaclgpu_<tag>.so: the WHOLE product library rebuilt under extra defines, objects in
# build_<tag>/ (for knobs that the host half shares with the kernels -- ACL_ROW_HASH: plan.cpp places the hashed rows with the hash the kernels probe with --
# where tools/build_variant.sh, which only rebuilds kernels.hip, would pair new kernels with old rows)
#        tools/build_variant_full.sh <tag> --rev <git rev>   -- the library as it was at <rev> (same-box A/B against an earlier round's kernels)
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$ROOT/spicedb-kubeapi-proxy_amd
SRC=$R/csrc
INC=$ROOT/include
if [ "$1" = "--rev" ]; then
  REV=$2; shift 2
  TMP=$(mktemp -d /tmp/aclgpu_rev_XXXX)
  git -C $ROOT archive $REV spicedb-kubeapi-proxy_amd/csrc include | tar -x -C $TMP
  SRC=$TMP/spicedb-kubeapi-proxy_amd/csrc
  INC=$TMP/include
fi
B=$R/build_$TAG
mkdir -p $B $R/lib
# (the sources find the ABI header as ../../include/aclgpu.h relative to csrc/: keep that shape for an archived revision)
pids=()
for f in $SRC/*.cpp $SRC/kernels.hip; do
  o=$B/$(basename $f).o
  if [[ $f == *.hip ]]; then X=""; else X="-x hip"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" $X -c $f -o $o &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/lib/libaclgpu_$TAG.so $B/*.o -ldl
echo built $R/lib/libaclgpu_$TAG.so
