# near-all-hit legs against gpucache_small_miss_insert_interval: bash tools/ab_small_insert.sh <tag>
TAG=${1:-r5ins}
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
for iv in 4 1 8 16; do
  BENCH_PS_EXTRA="{\"gpucache_small_miss_insert_interval\": $iv}" timeout 600 python bench.py --steps 20 --warmup 5 --blocks 2 --no-cpu-baseline --no-triton-leg --no-wide-leg --no-direct-leg --no-c3-leg --no-sharded-leg > gpurun_out/$TAG/iv${iv}_$rep.json 2> gpurun_out/$TAG/iv${iv}_$rep.err
  python - <<P
import json
d=json.loads(open("gpurun_out/$TAG/iv${iv}_$rep.json").read().strip().splitlines()[-1])
e=json.load(open("bench_extra.json"))["extra_legs"]
print("interval $iv run $rep: value %.3f G"%(d["value"]/1e9), " ".join("%s %.3f G (p50 %.2f ms, hit %.4f)"%(k[:7], e[k]["lookups_per_s"]/1e9, e[k]["p50_call_ms"], e[k]["measured_hit_rate"]) for k in ("all_hit_two_sessions_host_keys","hit_999_two_sessions_host_keys","hit_99_two_sessions_host_keys")))
P
done
done
