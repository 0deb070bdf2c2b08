# Generic A/B on the box: the headline region of bench.py under different environment switches, in the order given.
#   bash tools/ab_env.sh TAG "<extra bench.py arguments>" NAME1 "ENV1=.. ENV2=.. [-- <more arguments for this run>]" NAME2 "..." ...
TAG=$1; ARGS=$2; shift 2
mkdir -p gpurun_out/$TAG
while [ $# -ge 2 ]; do
  name=$1; envs=$2; extra=""; shift 2
  case "$envs" in *" -- "*) extra="${envs#* -- }"; envs="${envs%% -- *}";; esac   # "ENV=.. -- --more bench.py arguments"
  env $envs python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline $ARGS $extra > gpurun_out/$TAG/bench_$name.json 2> gpurun_out/$TAG/bench_$name.err
done
python - <<P
import json,glob,os
fs=sorted(glob.glob("gpurun_out/$TAG/bench_*.json"), key=os.path.getmtime)
for f in fs:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r=d["roofline"]
    print(f.split("bench_")[1][:-5], "%.3f G lookups/s"%(d["value"]/1e9), "frac", round(r["frac"],3), "probe %.1f gather %.1f scatter %.1f insert %.1f us"%(r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms"]*1e3), "ms/step %.3f"%d["ms_per_step"], "hit %.4f"%d.get("measured_hit_rate"), "p50 %.2f p99 %.2f ms"%(d["p50_batch_latency_ms"], d["p99_batch_latency_ms"]), "parity", d["parity_vs_oracle_bit_exact"], d.get("parity_full_batch"))
P
