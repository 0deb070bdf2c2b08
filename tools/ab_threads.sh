cat /sys/fs/cgroup/cpu.max; grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat
for th in 13 11 9; do
  HPS_SERVING_THREADS=$th python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('threads $th', 'value %.3fG p50 %.2f p99 %.2f blocks %s throttled %s' % (d['value']/1e9, d['p50_batch_latency_ms'], d['p99_batch_latency_ms'], [round(b,1) for b in d['block_ms']], d['host']))
"
  grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
done
