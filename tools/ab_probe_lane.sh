# K_P inside / outside the kernel lane: headline + near-all-hit legs.  bash tools/ab_probe_lane.sh <tag>
TAG=${1:-r5lane}
mkdir -p gpurun_out/$TAG
for rep in 1 2 3; do
for pl in ${PLS:-1 2}; do
  timeout 600 python bench.py --probe-in-lane $pl --steps 20 --warmup 5 --blocks 4 --no-cpu-baseline --no-triton-leg --no-wide-leg --no-direct-leg --no-c3-leg --no-sharded-leg > gpurun_out/$TAG/pl${pl}_$rep.json 2> gpurun_out/$TAG/pl${pl}_$rep.err
  python - <<P
import json
d=json.loads(open("gpurun_out/$TAG/pl${pl}_$rep.json").read().strip().splitlines()[-1])
e=json.load(open("bench_extra.json"))["extra_legs"]
r=d["roofline"]
print("probe_in_lane $pl run $rep: value %.3f G p50 %.2f frac %.3f probe %.1f gather %.1f |"%(d["value"]/1e9, d["p50_batch_latency_ms"], r["frac"], r["probe_ms"]*1e3, r["gather_ms"]*1e3), " ".join("%s %.3f G (p50 %.2f)"%(k.split("_two")[0].split("_one")[0], e[k]["lookups_per_s"]/1e9, e[k]["p50_call_ms"]) for k in ("device_keys", "all_hit_two_sessions_host_keys","hit_999_two_sessions_host_keys","hit_99_two_sessions_host_keys","hit_90_two_sessions_host_keys","hit_50_two_sessions_host_keys") if k in e))
P
done
done
