O=gpurun_out/triton_numa2; mkdir -p $O
for i in 1 2 3 4; do
for mode in free instbound; do
  E=""; [ $mode = instbound ] && E="HPS_BIND_INSTANCE_THREADS=1"
  env $E hugectr_backend_amd/lib/triton_abi_bench.bin --lib-dir hugectr_backend_amd/lib --tables 26 --rows 10000000 --dim 128 --batch 65536 \
      --cache-frac 0.2 --hit 0.957 --zipf 1.05 --instances 2 --steps 20 --blocks 12 --warmup 5 --direct 0 > $O/${mode}$i.json 2> $O/${mode}$i.err
  python3 - $O/${mode}$i.json $mode $i <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]) if l else {}
print(sys.argv[2],'run',sys.argv[3],'G %.3f'%(d.get('lookups_per_s',0)/1e9),'p50',d.get('p50_request_ms'),'p99',d.get('p99_request_ms'),'max',d.get('max_request_ms'),
      'blocks',[round(x,1) for x in d.get('block_ms',[])],'slow',d.get('slow_requests_ms'))
PY
done
done | tee $O/summary.txt
