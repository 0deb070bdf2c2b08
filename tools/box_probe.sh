# what kind of box is this?  gather-kernel time of the headline next to the GPU's partition modes and clocks
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -iE "partition" | head -4
rocm-smi --showclocks 2>/dev/null | grep -iE "mclk|sclk|fclk" | head -4
rocm-smi --showpower --showtemp 2>/dev/null | grep -iE "power|junction|memory" | head -5
python bench.py --steps 20 --warmup 5 --blocks 2 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.3f G gather %.1f us probe %.1f scatter %.1f insert %.1f frac %.3f d2d %.0f GB/s'%(d['value']/1e9, r['gather_ms']*1e3, r['probe_ms']*1e3, r['scatter_ms']*1e3, r['insert_ms']*1e3, r['frac'], r['box_d2d_copy_GBps']))"
rocm-smi --showclocks 2>/dev/null | grep -iE "mclk|sclk|fclk" | head -4
