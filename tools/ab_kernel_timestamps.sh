# The per-kernel times bench.py reports (engine option "timing") against rocprofv3's: kernels' own timestamps (default) and
# hipEventRecord pairs around the launches (HPS_KERNEL_TIMESTAMPS=0), same box, then the default under rocprofv3 --kernel-trace --stats
TAG=${1:-r3ts}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
line() { python -c "
import sys,json;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);r=d['roofline'];print('$2',round(d['value']/1e9,3),'frac',round(r['frac'],3),'probe %.1f gather %.1f scatter %.1f insert %.1f'%(r['probe_ms']*1e3,r['gather_ms']*1e3,r['scatter_ms']*1e3,r['insert_ms_not_counted']*1e3),d['parity_full_batch_vs_direct_row_index'])"; }
for i in 1 2; do
  HPS_KERNEL_TIMESTAMPS=1 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > $O/ts1_$i.json 2>/dev/null; line $O/ts1_$i.json "kernel timestamps   "
  HPS_KERNEL_TIMESTAMPS=0 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > $O/ts0_$i.json 2>/dev/null; line $O/ts0_$i.json "event pairs         "
done
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > $O/under_rocprof.json 2> $O/kt.log
cd $R; rm -f $O/kt/*kernel_trace.csv
line $O/under_rocprof.json "under rocprofv3     "
python - <<P
import csv,glob
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("probe_tile","gather_hits","miss_scatter","cache_insert")):
            print("rocprofv3 %-40s calls %s avg %.1f us min %.1f"%(r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
P
