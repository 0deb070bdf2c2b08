"""Diagnostic: latency of a session's calls by its miss-much mode, two sessions straddling the bound in opposite phase
(tests/test_gpu_lookup.py::test_per_call_switches_on_their_threshold_two_sessions_in_opposite_phase as a measurement)."""
import sys, threading
import numpy as np, torch
sys.path.insert(0, ".")
from tests.conftest import make_tables
from tests.test_gpu_lookup import _mk
from hugectr_backend_amd import hps

direct = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
T, R, D = 2, 200000, 64
tables = make_tables([(R, D)] * T, seed=61)
ps, cache, s0 = _mk("thr", tables, maxcat=[1] * T, gpucacheper=1.0, max_batch=90000, defaults=[2.5, -1.0], extra={"ps_direct_access": direct})
s1 = hps.LookupSession.create(ps, "thr", cache)
bound_rows = (1 << 20) // (D * 4)
nk = [85000, 85000]
res_idx = [np.nonzero(cache.query(t, tables[t][0]) >= 0)[0] for t in range(T)]
rec = [[], []]
def work(i, sess):
    sess.set_option("side_scatter_mb", 1)
    sess.set_option("timing", 1)
    rng = np.random.default_rng(100 + i)
    out = torch.empty(sum(nk) * D, dtype=torch.float32, device="cuda")
    for c in range(240):
        miss = int(bound_rows * (1.15 if (c + i) % 2 == 0 else 0.65))
        parts = []
        for t in range(T):
            idx = res_idx[t][rng.integers(0, res_idx[t].size, nk[t])]
            q = tables[t][0][idx].astype(np.int64)
            m = miss // T
            pos = rng.choice(nk[t], m, replace=False)
            q[pos] = -10 - (np.arange(m, dtype=np.int64) + (c * 4 + t) * 100000)
            parts.append(q)
        q = np.concatenate(parts)
        sess.lookup(q, nk, out=out)
        st = sess.last_stats()
        rec[i].append((int(st.miss_much_mode), miss, float(st.phase_ms[3]) + float(st.key_stage_ms), float(st.key_stage_ms), float(st.phase_ms[0]), float(st.phase_ms[1]), float(st.phase_ms[2]), float(st.gpu_call_ms)))
th = [threading.Thread(target=work, args=(0, s0)), threading.Thread(target=work, args=(1, s1))]
[t.start() for t in th]; [t.join() for t in th]
for i in range(2):
    a = np.array(rec[i][5:])
    for mode in (0, 1):
        for big in (True, False):
            sel = a[(a[:, 0] == mode) & ((a[:, 1] > bound_rows) == big)]
            if len(sel):
                print(f"session {i} mode {mode} missing {'1.15x' if big else '0.65x'}: n={len(sel)} call ms p50 {np.median(sel[:,2]):.3f} max {sel[:,2].max():.3f} | key stage {np.median(sel[:,3]):.3f} counts {np.median(sel[:,4]):.3f} fetch {np.median(sel[:,5]):.3f} tail {np.median(sel[:,6]):.3f} gpu span {np.median(sel[:,7]):.3f}")
    print("flips", s0.last_stats().mode_flips if i == 0 else s1.last_stats().mode_flips)
