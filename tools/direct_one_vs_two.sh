for a in "--direct 1 --sessions 1" "--direct 1 --sessions 2"; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-triton-leg $a 2>/dev/null | tail -1 | python3 -c '
import json, sys
d = json.loads(sys.stdin.read()); r = d["roofline"]; p = d["roofline_pcie"]
print("%-28s value %.3fG probe %.1f gather %.1f scatter %.1f insert %.1f fetch_ms %s pcie %.1f GB/s uniq %.0f phases %s" % (sys.argv[1], d["value"]/1e9, r["probe_ms"]*1e3, r["gather_ms"]*1e3, r["scatter_ms"]*1e3, r["insert_ms_not_counted"]*1e3, p.get("fetch_kernel_ms"), p["achieved"], p["unique_missed_rows_per_batch"], {k: round(v, 3) for k, v in d["mean_phase_ms"].items()}))' "$a"
done
