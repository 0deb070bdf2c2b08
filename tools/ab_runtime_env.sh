# Does a knob remove the 7-9 ms enqueue stalls of a fresh process' first seconds?  Each configuration several times.
L=hugectr_backend_amd/lib
for e in "X=0" "HPS_SYNC_COPY_STREAM=1" "X=0" "HPS_SYNC_COPY_STREAM=1" "X=0" "HPS_SYNC_COPY_STREAM=1" "X=0" "HPS_SYNC_COPY_STREAM=1" "DEBUG_CLR_MAX_BATCH_SIZE=16" "DEBUG_CLR_MAX_BATCH_SIZE=16" "ROC_AQL_QUEUE_SIZE=65536" "ROC_AQL_QUEUE_SIZE=65536"; do
  env $e timeout 600 $L/triton_abi_bench.bin --lib-dir $L --steps 20 --blocks 3 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', [round(b,1) for b in d['block_ms']], 'slow:', d['slow_requests_ms'])
"
done
