#!/bin/bash
# The plugin boundary with and without the serving process bound to the GPU's NUMA node (what INTEGRATION.md 4.3 recommends for
# tritonserver): N processes each way, interleaved.   bash tools/triton_numa_runs.sh [N=5] [tag]
N=${1:-5}; TAG=${2:-triton_numa}
O=gpurun_out/$TAG; mkdir -p $O
BUS=$(python - <<'P'
import ctypes as C
h=C.CDLL("/opt/rocm/lib/libamdhip64.so"); b=C.create_string_buffer(64)
assert h.hipDeviceGetPCIBusId(b, 64, 0)==0; print(b.value.decode().lower())
P
)
NODE=$(cat /sys/bus/pci/devices/$BUS/numa_node)
CPUS=$(cat /sys/devices/system/node/node$NODE/cpulist)
echo "GPU 0 at $BUS, NUMA node $NODE, cpus $CPUS" | tee $O/summary.txt
for i in $(seq 1 $N); do
for mode in free bound; do
  PRE=""; [ $mode = bound ] && PRE="taskset -c $CPUS"
  HPS_TRACE_TAIL=4 $PRE hugectr_backend_amd/lib/triton_abi_bench.bin --lib-dir hugectr_backend_amd/lib --tables 26 --rows 10000000 --dim 128 --batch 65536 \
      --cache-frac 0.2 --hit 0.957 --zipf 1.05 --instances 2 --steps 20 --blocks 12 --warmup 5 --direct 0 > $O/${mode}$i.json 2> $O/${mode}$i.err
  python3 - $O/${mode}$i.json $mode $i <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]) if l else {}
print(sys.argv[2],'run',sys.argv[3],'G %.3f'%(d.get('lookups_per_s',0)/1e9),'p50',d.get('p50_request_ms'),'p99',d.get('p99_request_ms'),'max',d.get('max_request_ms'),
      'blocks',[round(x,1) for x in d.get('block_ms',[])],'slow',d.get('slow_requests_ms'))
PY
done
done | tee -a $O/summary.txt
