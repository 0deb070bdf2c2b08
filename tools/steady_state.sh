# 1,200 timed steps of the headline workload (12 blocks of 100): do the block times stay flat once the cache has turned over?
# New defaults (turnover clock, insert age 2.5 turnovers, admission) against rounds 2-3's policy (8 calls per unit, 256 calls, no admission).
TAG=${1:-r3steady}
mkdir -p gpurun_out/$TAG
run() { n=$1; shift; env "$@" timeout 800 python bench.py --steps 100 --warmup 5 --blocks 12 --no-extra-legs --no-cpu-baseline > gpurun_out/$TAG/$n.json 2> gpurun_out/$TAG/$n.err
python - <<P
import json
d=json.loads(open("gpurun_out/$TAG/$n.json").read().strip().splitlines()[-1])
bl=d.get("block_ms") or d.get("blocks_ms") or []
print("$n", "%.3f G lookups/s"%(d["value"]/1e9), "hit %.4f"%d["measured_hit_rate"], "frac %.3f"%d["roofline"]["frac"], "blocks of 100 steps (ms)", [round(x,1) for x in bl], "parity", d["parity_vs_oracle_bit_exact"], d["parity_full_batch_vs_direct_row_index"])
P
}
run new HPS_X=1
run old HPS_LRU_AGE_SHIFT=3 HPS_LRU_ADMIT=0
