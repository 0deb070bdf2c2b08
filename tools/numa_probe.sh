# Is the headline's run-to-run spread NUMA placement?  bench under taskset: the GPU's node, the other node, no restriction.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
BUS=$(python - <<'P'
import ctypes as C
h=C.CDLL("/opt/rocm/lib/libamdhip64.so")
b=C.create_string_buffer(64)
assert h.hipDeviceGetPCIBusId(b, 64, 0)==0
print(b.value.decode().lower())
P
)
NODE=$(cat /sys/bus/pci/devices/$BUS/numa_node)
OTHER=$((1-NODE))
echo "GPU 0 at $BUS on NUMA node $NODE; cpus $(cat /sys/devices/system/node/node$NODE/cpulist) | other node cpus $(cat /sys/devices/system/node/node$OTHER/cpulist)"
for pass in 1 2; do
for cfg in local other free; do
  case $cfg in
    local) PRE="taskset -c $(cat /sys/devices/system/node/node$NODE/cpulist)";;
    other) PRE="taskset -c $(cat /sys/devices/system/node/node$OTHER/cpulist)";;
    free) PRE="";;
  esac
  $PRE timeout 400 python bench.py --steps 20 --warmup 5 --blocks 12 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python - <<P
import json
d=json.loads(open("/tmp/line.json").read()); e=json.load(open("bench_extra.json"))
m=e["mean_phase_ms"]
print("$cfg pass $pass: value %.3f G blocks %s pcie %.1f p50 %.2f | stage %.2f counts %.2f fetch %.2f tail %.2f"%(d["value"]/1e9,[round(x,1) for x in e["block_ms"]],d["roofline_pcie"]["achieved"],d["p50_batch_latency_ms"], e.get("key_stage_ms_mean",0), m["probe_until_counts_on_host"], m["ps_fetch"], m["h2d_scatter_insert"]))
P
done
done
