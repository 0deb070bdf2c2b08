for a in "--sessions 1" "--sessions 2" "--sessions 1 --split-probe 0" "--sessions 1 --hit 1.1"; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-triton-leg $a 2>/dev/null | tail -1 | python3 -c '
import json, sys
d = json.loads(sys.stdin.read()); r = d["roofline"]
print("%-32s value %.3fG probe %.1f gather %.1f scatter %.1f insert %.1f frac %.3f hit %.4f phases %s" % (sys.argv[1], d["value"]/1e9, r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms_not_counted"]*1e3, r["frac"], d["measured_hit_rate"], {k: round(v, 3) for k, v in d["mean_phase_ms"].items()}))' "$a"
done
