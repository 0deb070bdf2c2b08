#!/bin/bash
# Where does a fresh process get stuck in its first lookups?  Runs tools/hang_snippet.py under rocgdb N times; a run still there
# after 45 s is interrupted and every thread's backtrace printed.   bash tools/hang_repro_py.sh [N=30] [env assignments...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
N=${1:-30}; shift || true
for a in "$@"; do export "$a"; done
hung=0
for try in $(seq 1 $N); do
  /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex "handle SIGPIPE nostop noprint" -ex run -ex "thread apply all bt 24" --args python tools/hang_snippet.py > /tmp/gdb_out.txt 2>&1 &
  GP=$!
  for s in $(seq 1 45); do sleep 1; kill -0 $GP 2>/dev/null || break; done
  if kill -0 $GP 2>/dev/null; then
    C=$(pgrep -P $GP python | head -1)
    echo "== try $try: still running after 45 s: interrupting pid $C for stacks"
    kill -INT $C
    sleep 25
    grep -v "^\[New\|^\[Thread.*exited\|^warning\|Missing separate\|^Reading\|^Using host" /tmp/gdb_out.txt | cut -c1-260 | head -700
    kill -9 $C $GP 2>/dev/null
    hung=1
    break
  fi
  wait $GP
  echo "try $try: $(grep -c 'SNIPPET OK' /tmp/gdb_out.txt) ok $(grep -m1 'Error\|error' /tmp/gdb_out.txt | cut -c1-120)"
done
echo "hung=$hung"
