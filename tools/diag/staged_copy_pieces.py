"""Diagnostic: the staged_copy transport of the table-sharded entry session on the devices there are — request time by piece size
(P logical shards when there is one GPU).  python tools/diag/staged_copy_pieces.py [P] [rows_log2]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from hugectr_backend_amd import hps

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
ndev = torch.cuda.device_count()
N, D = 26 * 65536, 128
cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
       "models": [{"model": "m", "sparse_files": ["synthetic://x"], "num_of_worker_buffer_in_pool": P, "embedding_vecsize_per_table": [D],
                   "maxnum_catfeature_query_per_table_per_sample": [1], "default_value_for_each_table": [0.0],
                   "deployed_device_list": [s % ndev for s in range(P)], "max_batch_size": N, "gpucache": True, "gpucacheper": 1.0,
                   "gpucache_load_factor": 0.5, "hit_rate_threshold": 1.0, "table_sharding": "hash"}]}
ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
ps.load_table_synthetic("m", 0, 20260929, 0, rows)
ps.create_embedding_cache_per_model("m")
e = hps.ShardedEntrySession.create(ps, "m", 0)
rng = np.random.default_rng(1)
bt = [rng.integers(0, rows, N, dtype=np.int64) for _ in range(4)]
out = torch.empty(N * D, dtype=torch.float32, device="cuda:0")
def run(label, steps=20):
    for i in range(3):
        e.lookup(bt[i % 4], [N], out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ph = []
    for i in range(steps):
        e.lookup(bt[i % 4], [N], out=out)
        st = e.last_stats()
        ph.append((st.key_stage_ms, st.bucket_ms, st.lookup_ms, st.expand_ms, max(st.shard_ms[:P]), max(st.copy_wait_ms[:P]), st.passes[0]))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pm = np.mean(np.array(ph), axis=0)
    print(f"{label:28s} {N / dt / 1e9:6.3f} G lookups/s  request {dt * 1e3:6.3f} ms | key stage {pm[0]:.3f} bucket {pm[1]:.3f} shard lookups {pm[2]:.3f} (slowest shard {pm[4]:.3f}, waiting for copies {pm[5]:.3f}, pieces {int(pm[6])}) expand {pm[3]:.3f}", flush=True)
e.set_option("transport", 0); run("peer_store")
e.set_option("transport", 1)
for pk in (32768, 65536, 131072, 262144, 1 << 20, 0):
    e.set_option("copy_piece_keys", pk); run(f"staged_copy piece {pk if pk else 'automatic'}")
