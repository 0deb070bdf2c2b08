#!/bin/bash
# API trace (HIP runtime + HSA) of the native Triton-ABI driver, three processes; only the scan of slow calls is kept.
# usage (GPU box): tools/slow_api_trace.sh
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/slow_api_calls.txt
: > $out
L=hugectr_backend_amd/lib
for i in $(seq 1 ${1:-3}); do
  d=/tmp/apitrace_$i
  rm -rf $d
  timeout 900 rocprofv3 --hip-runtime-trace --hsa-core-trace --hsa-amd-trace --output-format csv -d $d -- \
    $L/triton_abi_bench.bin --lib-dir $L --tables 26 --rows 10000000 --dim 128 --batch 65536 --cache-frac 0.2 --hit 0.957 --zipf 1.05 \
    --instances 2 --steps 20 --blocks 6 --warmup 5 --direct 0 > /tmp/apitrace_$i.log 2>&1
  echo "=== process $i (rc $?) ===" >> $out
  grep -E "^\{" /tmp/apitrace_$i.log | python3 -c '
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print("  lookups/s %.3fG p50 %s p99 %s blocks %s slow requests [ms, watchdog gap] %s" % (d.get("lookups_per_s", 0) / 1e9, d.get("p50_request_ms"), d.get("p99_request_ms"), d.get("block_ms"), d.get("slow_requests_ms")))' >> $out 2>&1
  du -sh $d >> $out 2>&1
  python3 tools/slow_api_calls.py $d --ms 2.0 >> $out 2>&1
  rm -rf $d
done
cat $out
