#!/bin/bash
# The gather's duration as bench.py reports it (the kernels' own start/stop timestamps, hipExtLaunchKernel events) against rocprofv3's
# kernel trace OF THE SAME RUN, launch by launch.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/ktc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktc -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > /tmp/line.json 2>/dev/null
python - <<'PY'
import csv, glob, json
import numpy as np
f = glob.glob("/tmp/ktc/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "hps_gather_hits" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
full = json.load(open("/tmp/bench_extra.json"))
own = np.array(full["gather_us_per_call"], dtype=float)
print(f"rocprofv3: {len(dur)} launches, mean {dur.mean():.1f} us; launches 5..{5 + len(own) - 1} (the timed region): mean {dur[5:5 + len(own)].mean():.1f}, median {np.median(dur[5:5 + len(own)]):.1f}")
print(f"bench.py : {len(own)} calls, mean {own.mean():.1f} us, median {np.median(own):.1f}; roofline.gather_ms {full['roofline']['gather_ms'] * 1e3:.1f}")
a, b = np.sort(dur[5:5 + len(own)]), np.sort(own)
for q in (0, 10, 50, 90, 99, 100):
    print(f"  percentile {q:3d}: rocprofv3 {np.percentile(a, q):7.1f}   bench.py {np.percentile(b, q):7.1f}")
print("by block (probe, gather, scatter, insert us):", full["kernel_us_by_block"])
PY
