#!/bin/bash
# Run on the MI355X box from the repo root:  bash tools/collect_profile.sh <tag>
# Produces gpurun_out/<tag>/ — the round's ONE profile set, the files profiles/roundN/<tag>/ holds:
#   pytest_gpu.txt                          python -m pytest tests -m gpu
#   rocprofv3_kernel_stats_bench.csv        rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline`
#   bench_under_rocprofv3_kernel_trace.json the compact line of that profiled run
#   pmc_{FETCH,WRITE}_SIZE_counter_collection.csv   separate --pmc passes (one session), kernels of one lookup call
#   rocprof_summary.json                    tools/summarize_profile.py over the three
#   bench_default.json / bench_extra.json   the driver's command, un-profiled: compact line / full result
#   (rocprof_summary.json also splits hps_cache_insert_kernel's launches into cache warm-up and serving: insert_kernel_split)
# (kernel traces are deleted: only the small summaries travel back and get committed under profiles/).
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -2 $O/pytest_gpu.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline \
    > $O/bench_under_rocprofv3_kernel_trace.json 2> $O/kt.log
# (the per-launch trace stays until tools/summarize_profile.py has split the insert kernel's launches into warm-up and serving)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "hps_probe_tile|hps_miss_unique|hps_gather_hits|hps_miss_scatter|hps_cache_insert" --output-format csv -d $O/pmc_$C -o pmc -- \
      python $R/bench.py --steps 12 --warmup 4 --blocks 1 --sessions 1 --no-cpu-baseline --no-extra-legs > $O/pmc_$C.json 2> $O/pmc_$C.log
  find $O/pmc_$C -name "*kernel_trace.csv" -delete
  find $O/pmc_$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc_${C}_counter_collection.csv
done
cd $R
# summarize_profile.py looks for <dir>/kt/*kernel_stats.csv and <dir>/pmc_<C>/*counter_collection.csv
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kt/ 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do cp $O/pmc_${C}_counter_collection.csv $O/pmc_$C/ 2>/dev/null; done
python tools/summarize_profile.py $O $O/rocprof_summary.json > /dev/null
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_bench.csv
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/*.log $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cp bench_extra.json $O/bench_extra.json
tail -c 600 $O/bench_default.json
du -sh $O
