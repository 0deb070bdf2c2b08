"""Diagnostic: two entry sessions alternating 'repeats little' / 'repeats much' requests in opposite phase — the slow requests' phases."""
import sys, threading
import numpy as np, torch
sys.path.insert(0, ".")
from tests.conftest import make_tables
from tests.test_shard_entry import _server
from hugectr_backend_amd import hps
tables = make_tables([(300000, 16)], seed=13)
keys, rows = tables[0]
ps = _server("thr", tables, 2, gpucacheper=1.0, hit_rate_threshold=1.0, maxcat=[1], max_batch=70000,
                 extra={"gpucache_load_factor": 0.25})      # (every row resident: no miss path in the timings)
entries = [hps.ShardedEntrySession.create(ps, "thr", 0) for _ in range(2)]
N, calls = 66000, 200
rng = np.random.default_rng(5)
little = [rng.permutation(keys.size)[:N] for _ in range(4)]
much = [np.repeat(rng.permutation(keys.size)[:N // 2], 2) for _ in range(4)]
rec = [[], []]
def work(i):
    e = entries[i]
    out = torch.empty(N * 16, dtype=torch.float32, device="cuda")
    for c in range(calls):
        idx = (little if (c + i) % 2 == 0 else much)[c % 4]
        q = keys[idx].astype(np.int64)
        e.lookup(q, [N], out=out)
        st = e.last_stats()
        rec[i].append((c, (c + i) % 2, st.dedup_level, st.key_stage_ms, st.bucket_ms, st.lookup_ms, st.expand_ms, st.shard_ms[0], st.shard_ms[1], st.unique_keys, st.misses))
for rep in range(2):
    for r in rec: r.clear()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(2):
        a = np.array(rec[i], dtype=np.float64)
        tot = a[:, 3] + a[:, 4] + a[:, 5] + a[:, 6]
        print(f"pass {rep} session {i}: median {np.median(tot):.3f} max {tot.max():.3f}; by phase medians stage {np.median(a[:,3]):.3f} bucket {np.median(a[:,4]):.3f} lookups {np.median(a[:,5]):.3f} expand {np.median(a[:,6]):.3f}")
        for r, t in zip(rec[i], tot):
            if t > 0.5:
                print("   slow:", "call %d kind %d level %d stage %.3f bucket %.3f lookups %.3f expand %.3f shard0 %.3f shard1 %.3f unique %d misses %d" % r)
