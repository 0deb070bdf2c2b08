# A/B on the box: admission rule of the insert kernel (HPS_LRU_ADMIT: 0 = every new key takes the bucket's oldest slot,
# k = a slot hit more recently than the insert age is not given up, except for one new key in 2^k)
TAG=${1:-r3admit}
mkdir -p gpurun_out/$TAG
if [ "${2:-tests}" = tests ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.txt 2>&1; tail -3 gpurun_out/$TAG/pytest.txt
fi
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/$TAG/bench_$name.json 2> gpurun_out/$TAG/bench_$name.err; }
run A1_admit0 HPS_LRU_ADMIT=0
run B1_admit4 HPS_LRU_ADMIT=4
run A2_admit0 HPS_LRU_ADMIT=0
run B2_admit4 HPS_LRU_ADMIT=4
run C1_admit15 HPS_LRU_ADMIT=15
run D1_admit2 HPS_LRU_ADMIT=2
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/$TAG/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r=d["roofline"]
    print(f.split("bench_")[1][:-5], round(d["value"]/1e9,3), "frac", round(r["frac"],3), "probe %.1f gather %.1f scatter %.1f insert %.1f"%(r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms_not_counted"]*1e3), "hit %.4f"%d.get("measured_hit_rate"), "p50 %.2f p99 %.2f"%(d["p50_batch_latency_ms"], d["p99_batch_latency_ms"]), "parity", d["parity_vs_oracle_bit_exact"], d["parity_full_batch_vs_direct_row_index"], "blocks", d.get("block_ms", d.get("blocks_ms"))[:12] if (d.get("block_ms") or d.get("blocks_ms")) else "")
P
