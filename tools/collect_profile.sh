#!/bin/bash
# Run on the MI355X box from the repo root:  bash tools/collect_profile.sh <tag>
# Produces gpurun_out/<tag>/{bench_default.json, bench_under_rocprof.json, kernel_stats.csv, summary.json}
# (kernel traces are deleted: only the small summaries travel back and get committed under profiles/).
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline \
    > $O/bench_under_rocprof.json 2> $O/kt.log
rm -f $O/kt/kt_kernel_trace.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "hps_probe_tile|hps_miss_unique|hps_gather_hits|hps_miss_scatter|hps_cache_insert" --output-format csv -d $O/pmc_$C -o pmc -- \
      python $R/bench.py --steps 12 --warmup 4 --blocks 1 --sessions 1 --no-cpu-baseline --no-extra-legs > $O/pmc_$C.json 2> $O/pmc_$C.log
  rm -f $O/pmc_$C/pmc_kernel_trace.csv
done
cd $R
python tools/summarize_profile.py $O $O/summary.json > /dev/null
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
du -sh $O
