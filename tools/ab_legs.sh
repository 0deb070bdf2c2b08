#!/bin/bash
# On the MI355X box: the headline and the near-all-hit legs under different environment switches, interleaved.
#   bash tools/ab_legs.sh TAG ROUNDS NAME1 "ENV.." NAME2 "ENV.." ...
TAG=$1; ROUNDS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
for rep in $(seq 1 $ROUNDS); do
  set -- "$@"
  i=1
  while [ $i -lt $# ]; do
    name=${!i}; j=$((i+1)); envs=${!j}; i=$((i+2))
    env $envs python bench.py --steps 20 --warmup 5 --no-triton-leg --no-wide-leg --no-direct-leg --no-c3-leg --no-cpu-baseline > $O/${name}_$rep.out 2>/dev/null
    cp bench_extra.json $O/${name}_$rep.json
    python3 - $O/${name}_$rep.json ${name}_$rep <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); ex=d['extra_legs']; g=lambda k: ex[k]['lookups_per_s']/1e9
r=d['roofline']
print('%-14s headline %.3f G p50 %.3f p99 %.3f frac %.3f counts-on-host %.3f ms | all-hit %.2f  99.9%% %.2f  99%% %.2f  one-session p50 %.3f  dev-keys %.2f  call/kernel %.3f' % (sys.argv[2], d['value']/1e9, d['p50_batch_latency_ms'], d['p99_batch_latency_ms'], r['frac'], d['mean_phase_ms']['probe_until_counts_on_host'], g('all_hit_two_sessions_host_keys'), g('hit_999_two_sessions_host_keys'), g('hit_99_two_sessions_host_keys'), ex['one_session_host_keys_95']['p50_call_ms'], g('device_keys'), r['all_hit_call_over_kernel_time']))
PY
  done
done | tee $O/summary.txt
