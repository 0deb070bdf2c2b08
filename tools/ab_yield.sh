#!/bin/bash
# On the MI355X box: do the pool's yields (round 4) help or hurt the tail?  Interleaved runs of the headline region, 600 calls each.
TAG=${1:-ab_yield}; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --blocks 30 --no-extra-legs --no-cpu-baseline > $O/$name.out 2>/dev/null; cp bench_extra.json $O/$name.json; python3 - $O/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); sl=[c[0] for c in d['slowest_calls_ms']]
import numpy as np
print('%-12s %.3f G  p50 %.3f p99 %.3f  slowest %s  blocks>20ms %d of %d' % (sys.argv[2], d['value']/1e9, d['p50_batch_latency_ms'], d['p99_batch_latency_ms'], sl, sum(b>20 for b in d['block_ms']), len(d['block_ms'])))
PY
}
for rep in 1 2 3 4; do
  run yield_$rep X=1
  run noyield_$rep HPS_POOL_YIELD=0
done | tee $O/summary.txt
