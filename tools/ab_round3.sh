# A/B of the round-3 cache line layout on the box: recency unit (HPS_LRU_AGE_SHIFT), kernel serialisation lane (HPS_EXCLUSIVE_KERNELS)
# bash tools/ab_round3.sh <tag>
TAG=${1:-r3ab}
mkdir -p gpurun_out/$TAG
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/$TAG/bench_$name.json 2> gpurun_out/$TAG/bench_$name.err; }
run A1 X=1
run B_shift0 HPS_LRU_AGE_SHIFT=0
run C_shift1 HPS_LRU_AGE_SHIFT=1
run D_shift3 HPS_LRU_AGE_SHIFT=3
run E_nonexcl HPS_EXCLUSIVE_KERNELS=0
run A2 X=1
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/$TAG/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r=d["roofline"]
    print(f.split("bench_")[1][:-5], round(d["value"]/1e9,3), "frac", round(r["frac"],3), "probe %.1f gather %.1f scatter %.1f insert %.1f"%(r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms_not_counted"]*1e3), "hit %.4f"%d.get("measured_hit_rate"), "p50 %.2f p99 %.2f"%(d["p50_batch_latency_ms"], d["p99_batch_latency_ms"]), "blocks", [round(x,1) for x in d["block_ms"][::3]])
P
