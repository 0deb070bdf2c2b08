# Block times of the headline per commit, interleaved on ONE box: bash tools/bisect_blocks.sh "<sha> <sha> ... HEAD" [passes=2]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for pass in $(seq 1 ${2:-2}); do
for sha in $1; do
  if [ "$sha" = HEAD ]; then D=$R; else D=$R/_bisect/$sha; fi
  (cd $D && timeout 400 python bench.py --steps 20 --warmup 5 --blocks 12 --no-extra-legs --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 > /tmp/line.json; python - <<P
import json
try:
    d=json.loads(open("/tmp/line.json").read()); e=json.load(open("bench_extra.json"))
    m=e["mean_phase_ms"]; print("$sha pass $pass: value %.3f G  blocks %s  pcie %.1f GB/s p50 %.2f | stage %.2f counts %.2f fetch %.2f tail %.2f | %s"%(d["value"]/1e9, [round(x,1) for x in e["block_ms"]], d["roofline_pcie"]["achieved"], d["p50_batch_latency_ms"], e.get("key_stage_ms_mean",0), m["probe_until_counts_on_host"], m["ps_fetch"], m["h2d_scatter_insert"], {k:v for k,v in e["host"].items() if "numa" in k}))
except Exception as ex: print("$sha pass $pass: FAILED", ex)
P
  )
done
done
