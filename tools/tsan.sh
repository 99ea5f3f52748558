#!/bin/bash
# ThreadSanitizer over the host side of libaclgpu.so (no GPU needed: store-only engines).  Builds an instrumented copy of the library
# under /tmp/aclgpu_tsan (the kernels' object is reused uninstrumented), then runs
#   tools/store_stress.cpp   writers + readers + watch + snapshot patch / compaction self-checks + single checks, all at once
#                            (a second time with a thread that re-runs acl_load_bootstrap under everybody else)
#   tools/batcher_bench.cpp  the micro-batcher's wake-up tree and completion queue under 32-64 native threads (passes refused after 20 us)
#   tools/engine_stress.cpp  (built only: needs a GPU) every concurrent call shape of the seam at once, answers compared; run it on a GPU box
#                            as $T/engine_stress [seconds] -- set T=tools/bin/tsan before building so that it travels with gpurun
# usage: [T=dir] bash tools/tsan.sh [rounds]      prints the number of TSan reports (0 expected) per program
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/spicedb-kubeapi-proxy_amd
T=${T:-/tmp/aclgpu_tsan}
mkdir -p $T
T=$(cd $T && pwd)
make -C $P -j8 lib/libaclgpu.so > /dev/null
for f in schema store plan plan_reverse engine engine_shard engine_shard_native engine_callers engine_async engine_list bootstrap_yaml; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -Wno-option-ignored -x hip -c $P/csrc/$f.cpp -o $T/$f.o 2> /dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=thread -Wno-option-ignored -shared -o $T/libaclgpu.so $T/*.o $P/build/kernels.hip.o -ldl 2> /dev/null
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O1 -g -std=c++17 -fsanitize=thread $R/tools/store_stress.cpp -I$R/include -L$T -laclgpu -lpthread -Wl,-rpath,$T -o $T/store_stress
$CXX -O1 -g -std=c++17 -fsanitize=thread $R/tools/batcher_bench.cpp -I$R/include -L$T -laclgpu -lpthread -Wl,-rpath,$T -o $T/batcher_bench
$CXX -O1 -g -std=c++17 -fsanitize=thread $R/tools/engine_stress.cpp -I$R/include -L$T -laclgpu -lpthread -Wl,-rpath,'$ORIGIN' -o $T/engine_stress
export TSAN_OPTIONS="halt_on_error=0"
set +e
timeout 900 $T/store_stress ${1:-150} > $T/store_stress.out 2> $T/store_stress.err; rc1=$?
timeout 900 $T/store_stress ${1:-150} reload > $T/store_reload.out 2> $T/store_reload.err; rc3=$?
ACL_BATCHER_SIM_PASS_US=20 timeout 300 $T/store_stress ${1:-150} restart > $T/store_restart.out 2> $T/store_restart.err; rc4=$?
ACL_BATCHER_SIM_PASS_US=20 timeout 900 $T/batcher_bench 100 32 64 > $T/batcher_bench.out 2> $T/batcher_bench.err; rc2=$?
echo "store_stress rc=$rc1: $(cat $T/store_stress.out | tail -1); ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $T/store_stress.err)"
echo "store_stress with bootstrap reloads rc=$rc3: $(cat $T/store_reload.out | tail -1); ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $T/store_reload.err)"
echo "store_stress with batcher restarts rc=$rc4: $(cat $T/store_restart.out | tail -1); ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $T/store_restart.err)"
echo "batcher_bench rc=$rc2 ($(grep -c '"mode"' $T/batcher_bench.out) runs); ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $T/batcher_bench.err)"
grep -h "SUMMARY" $T/store_stress.err $T/store_reload.err $T/store_restart.err $T/batcher_bench.err | sort | uniq -c | sort -rn | head -20
