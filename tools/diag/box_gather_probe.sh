#!/bin/bash
# One box, three views of the same question — is the gather's time a property of the BOX or of the run?
#   1. tools/micro/random_rows.bin: random 512-B row reads (and read+write) over footprints 0.25..64 GB, no engine code at all;
#   2. tools/kbench.py: the engine's probe+gather on ONE session with device keys, all-hit and 95 % hit;
#   3. bench.py's timed region (two sessions, host keys, 95 % hit), short.
# Output: gpurun_out/box_gather_probe.txt (one block per call; collect over several boxes by hand).
set -u
out=gpurun_out/box_gather_probe.txt; mkdir -p gpurun_out
{
  echo "== $(date -u +%FT%TZ) $(rocm-smi --showserial 2>/dev/null | grep -i serial | head -1)"
  rocm-smi --showclocks 2>/dev/null | grep -E "mclk|sclk|fclk" | head -4
  rocm-smi --showtemp --showpower 2>/dev/null | grep -E "Temperature|Power" | head -6
  ./tools/micro/random_rows.bin | tail -7
  python tools/kbench.py --variants 1002 --hit 1.1 --rows 10000000 --rounds 2 2>&1 | grep -i variant
  python tools/kbench.py --variants 1002 --hit 0.95 --rows 10000000 --rounds 2 2>&1 | grep -i variant
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --blocks 8 --no-extra-legs --no-cpu-baseline > /tmp/o.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/o.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("bench: value %.3f G  gather %.1f us probe %.1f us frac %.3f" % (d["value"]/1e9, r["gather_ms"]*1e3, r["probe_ms"]*1e3, r["frac"]))
PY
  rocm-smi --showtemp --showpower 2>/dev/null | grep -E "Temperature|Power" | head -6
} 2>&1 | tee $out
