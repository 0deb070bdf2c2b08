#!/bin/bash
# On the MI355X box: the headline workload through TRITONBACKEND_ModelInstanceExecute, N processes in a row, with the engine's
# slow-call trace on; prints p50 / p99 / max per run and every traced call.   bash tools/triton_ten_runs.sh [runs=10] [tag]
N=${1:-10}; TAG=${2:-triton10}
O=gpurun_out/$TAG; mkdir -p $O
for i in $(seq 1 $N); do
  HPS_TRACE_TAIL=4 hugectr_backend_amd/lib/triton_abi_bench.bin --lib-dir hugectr_backend_amd/lib --tables 26 --rows 10000000 --dim 128 --batch 65536 \
      --cache-frac 0.2 --hit 0.957 --zipf 1.05 --instances 2 --steps 20 --blocks 12 --warmup 5 --direct 0 > $O/run$i.json 2> $O/run$i.err
  python3 - $O/run$i.json $i <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]) if l else {}
print('run',sys.argv[2],'G %.3f'%(d.get('lookups_per_s',0)/1e9),'p50',d.get('p50_request_ms'),'p99',d.get('p99_request_ms'),'max',d.get('max_request_ms'),
      'p99/p50 %.2f'%(d.get('p99_request_ms',0)/max(d.get('p50_request_ms',1),1e-9)),'slow',d.get('slow_requests_ms'),'late wakeups',d.get('watchdog_late_wakeups_over_2ms'))
PY
  grep -h "hps call\|hps tail" $O/run$i.err | head -20
done | tee $O/summary.txt
