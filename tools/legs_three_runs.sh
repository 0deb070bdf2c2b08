# N default runs in a row (in-process legs only): which legs have slow calls, and what the engine saw of them
# usage: bash tools/legs_three_runs.sh <tag> [N=3]
TAG=${1:-r3legs}
N=${2:-3}
mkdir -p gpurun_out/$TAG
for i in $(seq 1 $N); do
  timeout 600 python bench.py --steps 20 --warmup 5 --blocks 4 --no-cpu-baseline --no-triton-leg --no-wide-leg --no-direct-leg > gpurun_out/$TAG/run$i.json 2> gpurun_out/$TAG/run$i.err
done
python - <<P
import json
for i in range(1, $N + 1):
    try: d=json.loads(open("gpurun_out/$TAG/run%d.json"%i).read().strip().splitlines()[-1])
    except Exception as e: print(i,"FAILED",e); continue
    print("run",i,"value %.3f G"%(d["value"]/1e9),"frac %.3f"%d["roofline"]["frac"],"p99 %.2f"%d["p99_batch_latency_ms"])
    for k,v in d["extra_legs"].items():
        if isinstance(v,dict) and "max_call_ms" in v:
            print("   %-40s %.3f G p50 %.2f max %.2f slowest %s throttled %s steal %s"%(k, v["lookups_per_s"]/1e9, v["p50_call_ms"], v["max_call_ms"], v.get("slowest_call_ms"), v.get("cpu_quota_throttled_ms"), v.get("hypervisor_steal_ms")))
P
