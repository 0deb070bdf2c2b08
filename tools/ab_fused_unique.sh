# A/B on the box: call-wide unique misses in the probe kernel's tail (default) against the separate hps_miss_unique launch
TAG=${1:-r3fu}
mkdir -p gpurun_out/$TAG
timeout 600 python -m pytest tests/test_gpu_lookup.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_direct.py tests/test_gpu_dense.py -m gpu -x -q > gpurun_out/$TAG/pytest.txt 2>&1; tail -2 gpurun_out/$TAG/pytest.txt
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/$TAG/bench_$name.json 2> gpurun_out/$TAG/bench_$name.err; }
run A1_fused HPS_FUSED_UNIQUE=1
run B1_separate HPS_FUSED_UNIQUE=0
run A2_fused HPS_FUSED_UNIQUE=1
run B2_separate HPS_FUSED_UNIQUE=0
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/$TAG/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r=d["roofline"]
    print(f.split("bench_")[1][:-5], round(d["value"]/1e9,3), "frac", round(r["frac"],3), "probe %.1f gather %.1f scatter %.1f insert %.1f"%(r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms_not_counted"]*1e3), "hit %.4f"%d.get("measured_hit_rate"), "p50 %.2f p99 %.2f"%(d["p50_batch_latency_ms"], d["p99_batch_latency_ms"]), "parity", d["parity_vs_oracle_bit_exact"], d["parity_full_batch_vs_direct_row_index"], "counts on host %.3f"%d["mean_phase_ms"]["probe_until_counts_on_host"])
P
