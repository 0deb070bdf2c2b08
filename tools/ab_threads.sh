# A/B of host-side knobs through the environment: bash tools/ab_threads.sh
for env in "HPS_PIECE_MB=4" "HPS_PIECE_MB=8" "HPS_PIECE_MB=2" "HPS_PIECE_MB=16"; do
  env $env python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$env', 'value %.3fG p50 %.2f p99 %.2f pcie %.1f GB/s blocks %s' % (d['value']/1e9, d['p50_batch_latency_ms'], d['p99_batch_latency_ms'], d['roofline_pcie']['achieved'], [round(b,1) for b in d['block_ms']]))
"
done
