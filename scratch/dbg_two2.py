import sys, threading, numpy as np
sys.path.insert(0, '.')
from tests.conftest import make_tables, ps_config
from tests.test_gpu_lookup import _mk, _queries, _bits
from hugectr_backend_amd import hps
from oracle import hps_oracle as O

tables = make_tables([(6000, 128), (6000, 16)])
rowmap = {r.tobytes(): int(k) for k, r in zip(*tables[0])}
ps, cache, s0 = _mk("sharedx", tables, maxcat=[1, 1], gpucacheper=1.0, max_batch=4096)
s1 = hps.LookupSession.create(ps, "sharedx", cache)
co = O.COracle()
for k, r in tables: co.add_table_arrays(k, r)
lines = []
def worker(sess, sid, seed):
    rng = np.random.default_rng(seed)
    for it in range(10):
        nk = [4096, 4096]
        q = _queries(rng, tables, nk, miss_frac=0.05)
        out = sess.lookup(q, nk).cpu().numpy()
        st = sess.last_stats()
        ref = co.lookup(q, nk, [0.0, 0.0])
        g = out[:4096*128].reshape(4096,128); r = ref[:4096*128].reshape(4096,128)
        bad = np.nonzero((_bits(g) != _bits(r)).any(axis=1))[0]
        for i in bad:
            kind = "zeros" if (g[i]==0).all() else ("row_of_key %s" % rowmap.get(g[i].tobytes(), "GARBAGE"))
            nbad = int((_bits(g[i]) != _bits(r[i])).sum())
            lines.append(f"s{sid} it{it} pos{i} key{q[i]} expect_default={bool((r[i]==0).all())} got={kind} nbad_elems={nbad} misses={st.misses}")
th = [threading.Thread(target=worker, args=(s0, 0, 100)), threading.Thread(target=worker, args=(s1, 1, 200))]
[t.start() for t in th]; [t.join() for t in th]
print("\n".join(lines[:40])); print(len(lines), cache.counters())
