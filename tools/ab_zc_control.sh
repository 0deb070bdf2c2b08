#!/bin/bash
# A/B: control words of a call over the compute queue (pull/push kernels, HPS_ZC_CONTROL=1, default) or as SDMA copies (=0).
# Small W&D-shaped requests (both tiers, host and device keys), then the headline workload with one and two sessions.
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/ab_zc_control.txt
: > $out
for zc in 0 1 0 1; do
  for d in 0 1; do for k in 0 1; do
    echo -n "zc=$zc " >> $out
    HPS_ZC_CONTROL=$zc python3 tools/small_request_breakdown.py $d $k 2>/dev/null | tail -1 >> $out
  done; done
done
for zc in 0 1; do
  for a in "--sessions 1 --hit 1.1" "--sessions 2 --hit 1.1" "--sessions 2"; do
    HPS_ZC_CONTROL=$zc timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-triton-leg $a 2>/dev/null | tail -1 | python3 -c '
import json, sys
d = json.loads(sys.stdin.read()); r = d["roofline"]
print("zc=%s %-24s value %.3fG ms/step %.3f p50 %.3f p99 %.3f probe %.1f gather %.1f scatter %.1f frac %.3f parity %s phases %s" % (sys.argv[2], sys.argv[1], d["value"]/1e9, d["ms_per_step"], d["p50_batch_latency_ms"], d["p99_batch_latency_ms"], r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["frac"], d["parity_vs_oracle_bit_exact"], {k: round(v, 3) for k, v in d["mean_phase_ms"].items()}))' "$a" $zc >> $out
  done
done
cat $out
