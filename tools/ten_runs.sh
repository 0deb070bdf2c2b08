#!/bin/bash
# Ten consecutive runs of the driver's command (headline region only: the untimed legs are skipped) on one box; one line
# per run: value, p50/p99 of a batch, the blocks.  Round-1 review item 9 ("10 consecutive default runs with p99 <= 1.5 x p50").
# usage (GPU box): tools/ten_runs.sh [out-file]
out=${1:-gpurun_out/ten_runs.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-triton-leg 2>/dev/null | tail -1 |
  python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
p50, p99 = d["p50_batch_latency_ms"], d["p99_batch_latency_ms"]
h = d["host"]
print("run %s value %.3fG ms/step %.3f p50 %.2f p99 %.2f p99/p50 %.2f hit %.4f frac %.3f slowest %.1f steal %s ms cpus busy %.1f blocks %s" % (
      sys.argv[1], d["value"] / 1e9, d["ms_per_step"], p50, p99, p99 / p50, d["measured_hit_rate"], d["roofline"]["frac"],
      d["slowest_calls_ms"][0][0], h.get("hypervisor_steal_ms_in_timed_region"), h.get("process_cpus_busy_in_timed_region", 0),
      [round(b, 1) for b in d["block_ms"]]))' "$i" >> "$out"
done
cat "$out"
