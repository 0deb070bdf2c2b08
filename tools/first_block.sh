#!/bin/bash
# Where does the slow first block come from?  Three arrangements x 3 runs on one box (headline region only).
out=gpurun_out/first_block.txt
: > $out
one() {  # label, env, extra args
  for i in 1 2 3; do
    env $2 timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup ${3:-5} --no-extra-legs --no-cpu-baseline --no-triton-leg 2>/dev/null | tail -1 |
    python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
print("%-28s value %.3fG p50 %.2f p99 %.2f slowest %.1f blocks %s" % (sys.argv[1], d["value"] / 1e9, d["p50_batch_latency_ms"], d["p99_batch_latency_ms"],
      d["slowest_calls_ms"][0][0], [round(b, 1) for b in d["block_ms"]]))' "$1" >> $out
  done
}
one "baseline warmup 5" "X=1" 5
one "warmup 100" "X=1" 100
one "HPS_WARM_RUNTIME=64" "HPS_WARM_RUNTIME=64" 5
one "baseline warmup 5 again" "X=1" 5
cat $out
