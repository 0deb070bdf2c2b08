# staged keys by copy-engine copies (0) or by the pull kernel (1): headline + in-process legs.  bash tools/ab_keys_by_kernel.sh <tag>
TAG=${1:-r5kk}
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
for kk in ${KKS:-0 2}; do
  timeout 600 python bench.py --keys-by-kernel $kk --steps 20 --warmup 5 --blocks 12 --no-cpu-baseline --no-triton-leg --no-direct-leg --no-c3-leg --no-sharded-leg > gpurun_out/$TAG/kk${kk}_$rep.json 2> gpurun_out/$TAG/kk${kk}_$rep.err
  python - <<P
import json
d=json.loads(open("gpurun_out/$TAG/kk${kk}_$rep.json").read().strip().splitlines()[-1])
e=json.load(open("bench_extra.json")); x=e["extra_legs"]
print("keys_by_kernel $kk run $rep: value %.3f G p50 %.2f blocks %s |"%(d["value"]/1e9, d["p50_batch_latency_ms"], [round(b,1) for b in e["block_ms"]]), " ".join("%s %.3f G (p50 %.2f)"%(k.split("_two")[0].split("_one")[0][:22], x[k]["lookups_per_s"]/1e9, x[k]["p50_call_ms"]) for k in ("one_session_host_keys_95","all_hit_two_sessions_host_keys","hit_999_two_sessions_host_keys","hit_99_two_sessions_host_keys","pageable_host_keys_8_byte_staging","policy_threshold_0.9") if k in x), "| wide 8B %.3f G"%(x["wide_keys_95"]["lookups_per_s"]/1e9) if "wide_keys_95" in x else "")
P
done
done
