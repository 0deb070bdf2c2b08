for cfg in "1" "2" "3"; do set -- $cfg; S=$1
timeout 400 python bench.py --steps 80 --warmup 10 --sessions $S --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('S=$S value %.3e ms/step %.3f p50 %.2f p99 %.2f kern %.3f'%(r['value'],r['ms_per_step'],r['p50_batch_latency_ms'],r['p99_batch_latency_ms'],r['roofline']['avg_kernel_ms']), {k:round(v,2) for k,v in r['mean_phase_ms'].items()})
"
done
