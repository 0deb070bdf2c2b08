#!/bin/bash
# Where is the native stress driver stuck?  Runs tests/sanitize/abi_driver under rocgdb; a run that is still there well past its
# run time is interrupted and every thread's backtrace printed.   bash tools/hang_stacks.sh gpu_sharded
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
D=$R/hugectr_backend_amd/lib
MODE=${1:-gpu_sharded}
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -D__HIP_PLATFORM_AMD__ -Iinclude -Ihugectr_backend_amd/csrc -I/opt/rocm/include tests/sanitize/abi_driver.cpp -L$D -lhps_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$D -Wl,-rpath,/opt/rocm/lib -pthread -o /tmp/abi_driver || exit 1
for try in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "thread apply all bt 16" --args /tmp/abi_driver $MODE 4 > /tmp/gdb_out.txt 2>&1 &
  GP=$!
  sleep 30
  C=$(pgrep -P $GP abi_driver | head -1)
  if [ -n "$C" ]; then
    echo "== try $try: still running after 30 s: interrupting for stacks"
    kill -INT $C
    sleep 20
    grep -v "^\[New\|^\[Thread.*exited\|^warning\|Missing separate\|^Reading\|^Using host" /tmp/gdb_out.txt | cut -c1-230 | head -500
    kill -9 $C $GP 2>/dev/null
    exit 0
  fi
  wait $GP; echo "try $try: $(grep '^abi_driver' /tmp/gdb_out.txt | tail -1)"
done
