#!/bin/bash
# The list-level filters' parallel host code (csrc/engine_list.cpp, json_index.hpp, the interning pool it runs on) under ThreadSanitizer and under
# AddressSanitizer + UBSan, on the CPU (store-only engine: tools/list_stress.cpp).  Builds instrumented copies of libaclgpu.so under /tmp (the kernels' object is
# reused uninstrumented, as in tools/tsan.sh).  usage: bash tools/list_sanitize.sh [items] [threads] [rounds]   prints the reports found (0 expected)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/spicedb-kubeapi-proxy_amd
make -C $P -j8 lib/libaclgpu.so > /dev/null
CXX=/opt/rocm/lib/llvm/bin/clang++
for SAN in thread address,undefined; do
  T=/tmp/aclgpu_san_$(echo $SAN | tr , _)
  mkdir -p $T
  for f in schema store plan plan_reverse engine engine_shard engine_shard_native engine_callers engine_async engine_list bootstrap_yaml; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=$SAN -fno-sanitize=vptr -Wno-option-ignored -x hip -c $P/csrc/$f.cpp -o $T/$f.o 2> /dev/null &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=$SAN -Wno-option-ignored -shared -o $T/libaclgpu.so $T/*.o $P/build/kernels.hip.o -ldl 2> /dev/null
  $CXX -O1 -g -std=c++17 -fsanitize=$SAN $R/tools/list_stress.cpp -I$R/include -L$T -laclgpu -lpthread -Wl,-rpath,$T -o $T/list_stress
  set +e
  TSAN_OPTIONS="halt_on_error=0" ASAN_OPTIONS="detect_leaks=0" UBSAN_OPTIONS="print_stacktrace=1" timeout 900 $T/list_stress ${1:-4000} ${2:-3} ${3:-4} > $T/out.txt 2> $T/err.txt; rc=$?
  set -e
  echo "-fsanitize=$SAN rc=$rc: $(tail -1 $T/out.txt); reports: $(grep -c -E 'WARNING: ThreadSanitizer|ERROR: AddressSanitizer|runtime error:' $T/err.txt)"
  grep -h -E "SUMMARY|runtime error:" $T/err.txt | sort | uniq -c | sort -rn | head -8
done
