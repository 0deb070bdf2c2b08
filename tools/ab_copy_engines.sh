#!/bin/bash
# A/B of the SDMA engine wake-up at model load (csrc/cache/copy_engines.cpp): native Triton-ABI driver, interleaved
# processes; then the driver's bench command.  usage (GPU box): tools/ab_copy_engines.sh [pairs] [bench runs]
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/ab_copy_engines.txt
: > $out
L=hugectr_backend_amd/lib
run() {  # label, HPS_WAKE_COPY_ENGINES value
  HPS_WAKE_COPY_ENGINES=$2 HPS_TRACE_TAIL=1 timeout 600 $L/triton_abi_bench.bin --lib-dir $L --tables 26 --rows 10000000 --dim 128 --batch 65536 \
    --cache-frac 0.2 --hit 0.957 --zipf 1.05 --instances 2 --steps 20 --blocks 12 --warmup 5 --direct 0 > /tmp/ab.log 2> /tmp/ab.err
  grep -m1 "copy engines" /tmp/ab.err >> $out
  grep -E "^\{" /tmp/ab.log | python3 -c '
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print("%-10s %.3fG p50 %.2f p99 %.2f blocks %s slow [ms, watchdog gap] %s" % (sys.argv[1], d["lookups_per_s"] / 1e9, d["p50_request_ms"], d["p99_request_ms"],
          [round(b, 1) for b in d["block_ms"]], d["slow_requests_ms"]))' "$1" >> $out
}
for i in $(seq 1 ${1:-4}); do
  run "wake=0" 0
  run "wake=1" 1
done
for i in $(seq 1 ${2:-4}); do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-triton-leg 2>/dev/null | tail -1 |
  python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
print("bench.py   value %.3fG p50 %.2f p99 %.2f p99/p50 %.2f slowest %.1f blocks %s" % (d["value"] / 1e9, d["p50_batch_latency_ms"], d["p99_batch_latency_ms"],
      d["p99_batch_latency_ms"] / d["p50_batch_latency_ms"], d["slowest_calls_ms"][0][0], [round(b, 1) for b in d["block_ms"]]))' >> $out
done
cat $out
