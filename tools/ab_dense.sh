#!/bin/bash
# On the MI355X box: A/B of the config-5 dense step's interaction kernel (rocprofv3 per-kernel averages).  bash tools/ab_dense.sh [tag]
TAG=${1:-ab_dense}; O=$PWD/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
run() { name=$1; shift; cd /tmp; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o k -- python $R/tools/dense_bench.py > $O/$name.txt 2>/dev/null; cd $R
  echo "$name: $(tail -1 $O/$name.txt)"; python3 - $O/$name <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dense' in r['Name']: print('    %-40s calls %s avg %.1f us' % (r['Name'].split('(')[0][:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
  rm -rf $O/$name; }
for rep in 1 2; do
  run base_$rep X=1
  run plain_$rep HPS_DENSE_NT=0



done 2>&1 | tee $O/summary.txt
