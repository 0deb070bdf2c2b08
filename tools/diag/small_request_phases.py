"""Diagnostic: where a small (W&D-sized) request spends its time inside the engine: 28,672 keys, D = [1, 16], all keys resident."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from hugectr_backend_amd import hps
R = 1_000_000
cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
       "models": [{"model": "m", "sparse_files": ["synthetic://a", "synthetic://b"], "num_of_worker_buffer_in_pool": 2,
                   "embedding_vecsize_per_table": [1, 16], "maxnum_catfeature_query_per_table_per_sample": [2, 26],
                   "default_value_for_each_table": [0.0, 0.0], "deployed_device_list": [0], "max_batch_size": 1024,
                   "gpucache": True, "gpucacheper": 0.2, "hit_rate_threshold": 1.0}]}
ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
for t in range(2):
    ps.load_table_synthetic("m", t, 20260929, 0, R)
ps.create_embedding_cache_per_model("m")
cache = ps.get_embedding_cache("m", 0)
s = hps.LookupSession.create(ps, "m", cache)
rng = np.random.default_rng(0)
res = [np.arange(200000)[cache.query(t, np.arange(200000, dtype=np.int64)) >= 0] for t in range(2)]
nk = [2048, 26624]
out = torch.empty(2048 + 26624 * 16, dtype=torch.float32, device="cuda")
for hit, timing in ((1.0, 1), (1.0, 0), (0.99, 1), (0.99, 0), (0.9, 1), (0.9, 0)):
    s.set_option("timing", timing)
    rec = []
    for it in range(300):
        q = np.concatenate([np.where(rng.random(n) < hit, rng.choice(res[t], n), rng.integers(200000, R, n)) for t, n in enumerate(nk)]).astype(np.int64)
        t0 = time.perf_counter()
        s.lookup(q, nk, out=out)
        dt = (time.perf_counter() - t0) * 1e3
        st = s.last_stats()
        rec.append((dt, st.key_stage_ms, st.phase_ms[0], st.phase_ms[1], st.phase_ms[2], st.phase_ms[3], st.gpu_call_ms, st.probe_gather_ms, st.hit_gather_ms, st.scatter_ms))
    a = np.median(np.array(rec[50:]), axis=0)
    print(f"hit {hit} timing {timing}: python call {a[0]:.3f} ms | key staging {a[1]:.3f} | engine: counts on host {a[2]:.3f}, ps fetch {a[3]:.3f}, tail {a[4]:.3f}, whole {a[5]:.3f} | GPU span {a[6]:.3f} (probe {a[7]:.3f}, gather {a[8]:.3f}, scatter {a[9]:.3f})")
