import sys, time, os, numpy as np
sys.path.insert(0,'.')
from hugectr_backend_amd import hps
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip(), '|', open('/sys/kernel/mm/transparent_hugepage/defrag').read().strip())
T,R,D=4,10_000_000,128
cfg={"supportlonglong":True,"volatile_db":{"type":"hash_map","num_partitions":8},"models":[{"model":"m","sparse_files":[f"s{t}" for t in range(T)],"num_of_worker_buffer_in_pool":2,"embedding_vecsize_per_table":[D]*T,"maxnum_catfeature_query_per_table_per_sample":[1]*T,"default_value_for_each_table":[0.0]*T,"deployed_device_list":[0],"max_batch_size":65536,"gpucache":False}]}
t0=time.time(); ps=hps.HierParameterServer.create_from_dict(cfg,load_tables=False)
for t in range(T): ps.load_table_synthetic("m",t,1,0,R)
print("load %.1fs"%(time.time()-t0))
for l in open('/proc/self/smaps_rollup'):
    if 'AnonHuge' in l or 'Rss' in l: print(l.strip())
rng=np.random.default_rng(0)
s=hps.LookupSession.create(ps,"m",None)
for n in (3277*T, 85000):
    per=[n//T]*T
    out=np.empty(sum(per)*D,np.float32)
    ts=[]
    for it in range(30):
        q=rng.integers(0,R,sum(per)).astype(np.int64)
        t0=time.perf_counter(); s.lookup(q,per,out=out); ts.append((time.perf_counter()-t0)*1e3)
    ts=np.array(ts[5:]); print(f"host-tier lookup of {sum(per)} keys over {T} tables: median {np.median(ts):.3f} ms min {ts.min():.3f} max {ts.max():.3f}  -> {sum(per)/np.median(ts)/1e3:.1f} M keys/s")
