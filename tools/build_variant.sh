#!/bin/bash
# usage: tools/build_variant.sh <tag> [-DNAME=VALUE ...] -- lib/libaclgpu_<tag>.so: the product library with kernels.hip rebuilt under extra defines
# (A/B of kernel variants on the GPU box: ACLGPU_LIB=.../libaclgpu_<tag>.so python bench.py ...; tools/ab.sh runs every variant found)
set -e
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)/spicedb-kubeapi-proxy_amd
make -s -C $R -j8 lib/libaclgpu.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $R/csrc/kernels.hip -o $R/build/kernels_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/lib/libaclgpu_$TAG.so $(ls $R/build/*.cpp.o) $R/build/kernels_$TAG.o -ldl
echo built $R/lib/libaclgpu_$TAG.so
