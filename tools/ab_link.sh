#!/bin/bash
# On the MI355X box: link-efficiency A/B of the headline workload (upload piece size, sessions), interleaved pairs.
# bash tools/ab_link.sh [tag]
TAG=${1:-ab_link}; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline $EXTRA > $O/$name.out 2>/dev/null; cp bench_extra.json $O/$name.json; python3 - $O/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; m=d['mean_phase_ms']
print('%-22s %.3f G  ms/step %.3f  p50 %.3f p99 %.3f  pcie %.3f (%.1f GB/s)  hit %.4f  fetch %.3f tail %.3f stage %.3f' % (sys.argv[2], d['value']/1e9, d['ms_per_step'], d['p50_batch_latency_ms'], d['p99_batch_latency_ms'], d['roofline_pcie']['frac'], d['roofline_pcie']['achieved'], d['measured_hit_rate'], m['ps_fetch'], m['h2d_scatter_insert'], d['key_stage_ms_mean']))
PY
}
for rep in 1 2; do
  EXTRA="" run default_$rep X=1
  EXTRA="" run piece8_$rep HPS_PIECE_MB=8
  EXTRA="" run piece16_$rep HPS_PIECE_MB=16
  EXTRA="--sessions 3" run sessions3_$rep X=1
  EXTRA="--sessions 3" run sessions3_piece8_$rep HPS_PIECE_MB=8
done | tee $O/summary.txt
