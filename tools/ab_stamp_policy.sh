#!/bin/bash
# LRU stamp policy A/B on the headline workload: sampled (1 hit in N), every hit, or age-checked (only stamps at least K calls old).
# usage: tools/ab_stamp_policy.sh [rounds] ["arg sets" ...]
cd "$(dirname "$0")/.." || exit 1
rounds=${1:-3}; shift
sets=("$@"); [ ${#sets[@]} -eq 0 ] && sets=("--stamp-every 4" "--stamp-every 1" "--stamp-stale 4")
for i in $(seq 1 $rounds); do
for a in "${sets[@]}"; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-triton-leg $a 2>/dev/null | tail -1 | python3 -c '
import json, sys
d = json.loads(sys.stdin.read()); r = d["roofline"]
print("%-18s value %.3fG ms/step %.3f hit %.4f probe %.1f gather %.1f scatter %.1f frac %.3f parity %s blocks %s" % (sys.argv[1], d["value"]/1e9, d["ms_per_step"], d["measured_hit_rate"], r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["frac"], d["parity_vs_oracle_bit_exact"], [round(b, 1) for b in d["block_ms"]]))' "$a"
done
done
