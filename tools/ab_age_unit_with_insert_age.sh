mkdir -p gpurun_out/r3age2
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r3age2/$name.json 2>/dev/null; python -c "
import json;d=json.loads([l for l in open('gpurun_out/r3age2/$name.json') if l.startswith('{')][-1]);r=d['roofline'];print('$name',round(d['value']/1e9,3),'hit %.4f'%d['measured_hit_rate'],'frac',round(r['frac'],3),'blocks',[round(x,1) for x in d['block_ms'][::3]], d['cache_counters']['dropped'])"; }
run s2_a64 X=1
run s3_a64 HPS_LRU_AGE_SHIFT=3
run s3_a32 HPS_LRU_AGE_SHIFT=3 HPS_LRU_INSERT_AGE=32
run s4_a16 HPS_LRU_AGE_SHIFT=4 HPS_LRU_INSERT_AGE=16
run s1_a128 HPS_LRU_AGE_SHIFT=1 HPS_LRU_INSERT_AGE=128
run s2_a64b X=1
