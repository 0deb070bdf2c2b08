#!/bin/bash
# Write-path counters of the config-5 interaction kernel (review of round 4, item 5): separate rocprofv3 --pmc passes over
# tests/test_gpu_dense.py::test_dense_interaction_at_the_full_config5_batch (65,536 samples x 26 x 128), kernel-filtered.
# Usage (GPU box): bash tools/dense_pmc.sh gpurun_out/<dir>
O=${1:-gpurun_out/dense_pmc}
mkdir -p $O
export TMPDIR=/tmp
for C in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_STALL_sum"; do
  tag=$(echo $C | tr ' ' '+')
  timeout 400 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "hps_dense_interact" --output-format csv -d $O/pmc_$tag -o pmc -- \
    python -m pytest tests/test_gpu_dense.py::test_dense_interaction_at_the_full_config5_batch -q -m gpu > $O/pmc_$tag.log 2>&1
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k}: launches {len(v)}  mean per launch {sum(v)/len(v):.6g}")
PY
done
timeout 400 rocprofv3 --kernel-trace --stats --kernel-include-regex "hps_dense" --output-format csv -d $O/trace -o t -- \
  python -m pytest tests/test_gpu_dense.py::test_dense_interaction_at_the_full_config5_batch -q -m gpu > $O/trace.log 2>&1
cut -c1-200 $(find $O/trace -name "*kernel_stats.csv" | head -1) | head -5
