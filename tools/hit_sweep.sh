# Hit-rate sweep of the headline workload: bash tools/hit_sweep.sh > gpurun_out/hit_sweep.txt
for h in 1.1 0.99 0.957 0.90 0.52; do
  python bench.py --steps 20 --warmup 5 --blocks 5 --no-extra-legs --no-cpu-baseline --hit $h 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('target hit $h: measured %.4f  value %.3f G lookups/s  ms/step %.3f  p50 %.2f ms  frac %.3f  probe %.1f gather %.1f scatter %.1f us  pcie %.1f GB/s  parity %s' % (d['measured_hit_rate'], d['value']/1e9, d['ms_per_step'], d['p50_batch_latency_ms'], r['frac'], r['probe_ms']*1e3, r['gather_ms']*1e3, r['scatter_ms']*1e3, d['roofline_pcie']['achieved'], d['parity_vs_oracle_bit_exact']))
"
done
