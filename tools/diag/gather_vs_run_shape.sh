run() { # label, args...
  L=$1; shift
  timeout 400 "$@" > /tmp/o.json 2>/dev/null
  python - "$L" <<PY
import json,sys
d=json.loads(open("/tmp/o.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("%-34s value %.3f G  gather %.1f probe %.1f scatter %.1f insert %.1f frac %.3f" % (sys.argv[1], d["value"]/1e9, r["gather_ms"]*1e3, r["probe_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms"]*1e3, r["frac"]))
PY
}
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline"
run "blocks 8" $B --blocks 8
run "blocks 12" $B --blocks 12
run "blocks 24" $B --blocks 24
run "blocks 8 again" $B --blocks 8
cd /tmp; export TMPDIR=/tmp
run "blocks 12 from /tmp" python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline
run "blocks 12 under rocprofv3" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python /root/repo/bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline
cd /root/repo
run "blocks 8 after" $B --blocks 8
