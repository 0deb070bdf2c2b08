#!/bin/bash
# On the GPU box: ThreadSanitizer build of the product libraries + the native ABI stress driver (tests/sanitize/abi_driver.cpp),
# run against the host tier, the GPU cache with host gather, the GPU cache with the device-driven tier, and a table-sharded model
# served by entry sessions.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/tsan_shell
mkdir -p $D
export HPS_AMD_LIB_DIR=$D HPS_AMD_EXTRA_FLAGS="-fsanitize=thread -fno-omit-frame-pointer -g -shared-libsan" HPS_AMD_EXTRA_LDFLAGS="-fsanitize=thread -shared-libsan"
cd $R && python -m hugectr_backend_amd.build > $D/build.log 2>&1 || { tail -5 $D/build.log; exit 1; }
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname $($CLANG -print-file-name=libclang_rt.tsan-x86_64.so))
$CLANG -std=c++17 -O1 -g -fsanitize=thread -shared-libsan -D__HIP_PLATFORM_AMD__ -Iinclude -Ihugectr_backend_amd/csrc -I/opt/rocm/include \
  tests/sanitize/abi_driver.cpp -L$D -lhps_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$D -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT -pthread -o $D/abi_driver || exit 1
for mode in cpu gpu gpu_direct gpu_sharded; do
  TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0" timeout 300 $D/abi_driver $mode ${1:-8} > $D/$mode.log 2>&1
  echo "$(grep '^abi_driver' $D/$mode.log | tail -1)  | ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' $D/$mode.log)"
  python tests/sanitize/tsan_filter.py $D/$mode.log | tail -12
done
