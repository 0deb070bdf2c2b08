# A/B runs of bench.py on the box: bash tools/ab_sessions.sh  (prints one line per configuration)
for cfg in "--chain-gather 0" "--chain-gather 1" "--chain-gather 0" "--chain-gather 1"; do
  python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg', 'value %.3fG ms/step %.3f p50 %.2f p99 %.2f hit %.4f frac %.3f probe %.1f gather %.1f scatter %.1f insert %.1f blocks %s' % (d['value']/1e9, d['ms_per_step'], d['p50_batch_latency_ms'], d['p99_batch_latency_ms'], d['measured_hit_rate'], r['frac'], r['probe_ms']*1e3, r['gather_ms']*1e3, r['scatter_ms']*1e3, r['insert_ms_not_counted']*1e3, [round(b,1) for b in d['block_ms']]))
"
done
