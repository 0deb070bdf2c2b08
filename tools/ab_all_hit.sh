# every key resident (--hit 1.1), two sessions, host keys: what bounds the call when nothing crosses PCIe but the keys
mkdir -p gpurun_out/allhit
run() { name=$1; shift; env "$@" python bench.py --hit 1.1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/allhit/$name.json 2>/dev/null; python -c "
import json;d=json.loads([l for l in open('gpurun_out/allhit/$name.json') if l.startswith('{')][-1]);r=d['roofline'];print('%-22s'%'$name',round(d['value']/1e9,3),'G  ms/step %.3f'%d['ms_per_step'],'p50 %.3f'%d['p50_batch_latency_ms'],'probe %.1f gather %.1f'%(r['probe_ms']*1e3,r['gather_ms']*1e3),'stage %.3f'%d['key_stage_ms_mean'],'phases',{k:round(v,3) for k,v in d['mean_phase_ms'].items()})"; }
run default X=1
run no_lane HPS_EXCLUSIVE_KERNELS=0
run event_pairs HPS_KERNEL_TIMESTAMPS=0
run separate_unique HPS_FUSED_UNIQUE=0
python bench.py --hit 1.1 --sessions 3 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]);print('three sessions        ',round(d['value']/1e9,3),'G  ms/step %.3f'%d['ms_per_step'],'p50 %.3f'%d['p50_batch_latency_ms'])"
run default_again X=1
