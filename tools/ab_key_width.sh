#!/bin/bash
# Keys over PCIe at 4 bytes (uint32) or 3 bytes (packed) on the headline workload, interleaved.
cd "$(dirname "$0")/.." || exit 1
for i in 1 2 3; do
for m in 2 1; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-triton-leg --narrow-keys $m 2>/dev/null | tail -1 | python3 -c '
import json, sys
d = json.loads(sys.stdin.read()); r = d["roofline"]
print("narrow_keys %s key bytes %.1f value %.3fG ms/step %.3f p50 %.2f p99 %.2f key stage %.3f ms probe %.1f frac %.3f pcie %.1f GB/s parity %s" % (sys.argv[1], d["key_bytes_over_pcie_mean"], d["value"]/1e9, d["ms_per_step"], d["p50_batch_latency_ms"], d["p99_batch_latency_ms"], d["key_stage_ms_mean"], r["probe_ms"]*1e3, r["frac"], d["roofline_pcie"]["achieved"], d["parity_vs_oracle_bit_exact"]))' $m
done
done
