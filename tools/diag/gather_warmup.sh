#!/bin/bash
# Does the gather's time depend on how long the GPU has been under load?  One fresh box: short bench x2, a burn of random-row traffic,
# short bench x2 — with clocks / power / temperature sampled every 2 s into gpurun_out/gather_warmup_smi.txt.
mkdir -p gpurun_out
S=gpurun_out/gather_warmup_smi.txt; : > $S
( while true; do echo "t=$(date +%s)" >> $S; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power \(W\)|junction|memory\)" | sed 's/GPU\[0\]\t\t: //' >> $S; sleep 2; done ) &
SP=$!
run() {
  L=$1; shift
  T0=$(date +%s)
  timeout 400 "$@" > /tmp/o.json 2>/dev/null
  python - "$L" $T0 <<PY
import json,sys,time
d=json.loads(open("/tmp/o.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("%-28s start %s end %d  value %.3f G  gather %.1f probe %.1f scatter %.1f insert %.1f frac %.3f" % (sys.argv[1], sys.argv[2], time.time(), d["value"]/1e9, r["gather_ms"]*1e3, r["probe_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms"]*1e3, r["frac"]))
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --blocks 8"
run "bench 1 (fresh box)" $B
run "bench 2" $B
echo "burn start $(date +%s)"
for i in 1 2 3 4 5 6; do ./tools/micro/random_rows.bin > /dev/null 2>&1; done
echo "burn end $(date +%s)"
run "bench 3 (after the burn)" $B
run "bench 4" $B
kill $SP
