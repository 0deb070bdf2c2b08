#!/bin/bash
# A/B of the parameter-server miss path on the GPU box: host gather vs ps_direct_access, over session counts
# and fetch-kernel grid sizes.  Prints one line per run; full JSON lines go to gpurun_out/ab_direct/.
# usage: bash tools/ab_direct.sh "<name>|<env>|<bench args>" ...
mkdir -p gpurun_out/ab_direct
run() {  # name, env, args...
  local name=$1; local envs=$2; shift 2
  env $envs timeout 500 python bench.py --no-cpu-baseline --no-extra-legs "$@" > gpurun_out/ab_direct/$name.json 2> gpurun_out/ab_direct/$name.err
  python - "$name" <<'EOF'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/ab_direct/{name}.json").read().strip().splitlines()[-1])
    print(f"{name:28s} {d['value']/1e9:6.3f} G/s  {d['ms_per_step']:.3f} ms/step  p50 {d['p50_batch_latency_ms']:.2f}  "
          f"p99 {d['p99_batch_latency_ms']:.2f}  thr {(d.get('host') or {}).get('cpu_quota_throttled_periods_in_timed_region')}  K_A {d['roofline']['avg_kernel_ms']*1e3:.0f} us  parity {d['parity_vs_oracle_bit_exact']}")
except Exception as e:
    print(name, "FAILED", e)
EOF
}
for spec in "$@"; do
  IFS='|' read -r name envs args <<< "$spec"
  run "$name" "${envs:-A=1}" $args
done
