# A/B on the box: stamp of newly inserted keys (HPS_LRU_INSERT_AGE, recency units in the past; 0 = plain LRU insertion)
TAG=${1:-r3ins}
mkdir -p gpurun_out/$TAG
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/$TAG/bench_$name.json 2> gpurun_out/$TAG/bench_$name.err; }
run A1_age0 HPS_LRU_INSERT_AGE=0
run B_age16 HPS_LRU_INSERT_AGE=16
run C_age48 HPS_LRU_INSERT_AGE=48
run D_age96 HPS_LRU_INSERT_AGE=96
run E_age160 HPS_LRU_INSERT_AGE=160
run A2_age0 HPS_LRU_INSERT_AGE=0
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/$TAG/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r=d["roofline"]
    print(f.split("bench_")[1][:-5], round(d["value"]/1e9,3), "frac", round(r["frac"],3), "probe %.1f gather %.1f scatter %.1f insert %.1f"%(r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms_not_counted"]*1e3), "hit %.4f"%d.get("measured_hit_rate"), "p50 %.2f p99 %.2f"%(d["p50_batch_latency_ms"], d["p99_batch_latency_ms"]), "parity", d["parity_full_batch_vs_direct_row_index"], "blocks", [round(x,1) for x in d["block_ms"][::3]])
P
