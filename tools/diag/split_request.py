"""Diagnostic: ONE request in flight, served (a) by one lookup session, (b) split by tables over two / four sessions of the same cache
driven side by side (each sub-call names the other tables with NUMKEYS 0), against (c) two independent sessions with one request each
(the headline's arrangement).  Config 2's call shape: 26 tables x 65,536 keys x 128 fp32, ~95.7 % hit, host keys, synchronous insert.
    python tools/diag/split_request.py [rows_per_table=2000000]
Environment: HIT=<resident draw probability, default 0.957>, NB=<distinct batches generated, default 600: every request must be a fresh one>,
ONLY_TWO=1 (three readings of one-in-flight / two-in-flight only — for alternating two builds of the library via HPS_AMD_LIB_DIR)."""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
from hugectr_backend_amd import hps

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
import os
T, B, D, HIT = 26, 65536, 128, float(os.environ.get("HIT", "0.957"))
C = int(R * 0.2)
cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
       "models": [{"model": "m", "sparse_files": [f"synthetic://t{t}" for t in range(T)], "num_of_worker_buffer_in_pool": 7,
                   "embedding_vecsize_per_table": [D] * T, "maxnum_catfeature_query_per_table_per_sample": [1] * T,
                   "default_value_for_each_table": [0.0] * T, "deployed_device_list": [0], "max_batch_size": B,
                   "gpucache": True, "gpucacheper": 0.2, "hit_rate_threshold": 1.0}]}
ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
for t in range(T):
    ps.load_table_synthetic("m", t, 20260930 + t, 0, R)
ps.create_embedding_cache_per_model("m")
cache = ps.get_embedding_cache("m", 0)
S = [hps.LookupSession.create(ps, "m", cache) for _ in range(7)]   # 0: whole request; 1-2: halves; 3-6: quarters (1-2 also: independent)
rng = np.random.default_rng(7)
res = [np.arange(C, dtype=np.int64)[cache.query(t, np.arange(C, dtype=np.int64)) >= 0] for t in range(T)]
def batch():
    q = np.empty(T * B, dtype=np.int64)
    for t in range(T):
        hit = rng.random(B) < HIT
        k = rng.integers(C, R, B)
        k[hit] = res[t][rng.integers(0, res[t].size, int(hit.sum()))]
        q[t * B:(t + 1) * B] = k
    return q
NB = int(os.environ.get('NB', '600'))
batches = [batch() for _ in range(NB)]   # every request of the run is a fresh batch (a repeated one would find its misses cached)
next_b = [0]
def fresh(n):
    b0 = next_b[0]; next_b[0] += n
    return [batches[(b0 + i) % NB] for i in range(n)]
outs = [torch.empty(T * B * D, dtype=torch.float32, device="cuda") for _ in range(2)]
P = hps.LookupSession
def args(q, out, tables):
    kp = P.pack_ptrs([q.ctypes.data + 8 * t * B for t in range(T)])
    vp = P.pack_ptrs([out.data_ptr() + 4 * t * B * D for t in range(T)])
    nk = P.pack_counts([B if t in tables else 0 for t in range(T)])
    return kp, vp, nk
ALL = set(range(T))
def parts(k):
    return [set(range(T * i // k, T * (i + 1) // k)) for i in range(k)]

class Pool:
    """persistent threads, one per sub-session; run(jobs) returns when every job's lookup has returned"""
    def __init__(self, n):
        self.n, self.jobs, self.go, self.done = n, [None] * n, [threading.Event() for _ in range(n)], [threading.Event() for _ in range(n)]
        self.th = [threading.Thread(target=self.work, args=(i,), daemon=True) for i in range(n)]
        [t.start() for t in self.th]
    def work(self, i):
        while True:
            self.go[i].wait(); self.go[i].clear()
            s, a = self.jobs[i]
            if i and STAGGER_US:   # sub-call i starts i x STAGGER_US after the first (out of phase on purpose)
                t_end = time.perf_counter() + i * STAGGER_US * 1e-6
                while time.perf_counter() < t_end: pass
            s.lookup_packed(*a)
            self.done[i].set()
    def run(self, jobs):
        for i, j in enumerate(jobs):
            self.jobs[i] = j; self.go[i].set()
        for i in range(len(jobs)):
            self.done[i].wait(); self.done[i].clear()
pool = Pool(4)
STAGGER_US = 0

def one_in_flight(k, steps):
    """k = 1: the whole request on session 0; k = 2 / 4: split by tables over sessions 1-2 / 3-6"""
    sess = [S[0]] if k == 1 else (S[1:3] if k == 2 else S[3:7])
    pre = [[(sess[i], args(q, outs[0], p)) for i, p in enumerate(parts(k))] for q in fresh(steps + 4)]
    lat = []
    for it in range(steps + 4):
        t0 = time.perf_counter()
        if k == 1:
            s, a = pre[it][0]; s.lookup_packed(*a)
        else:
            pool.run(pre[it])
        if it >= 4: lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat)
    global last_hit
    last_hit = 1.0 - sum(x.last_stats().misses for x in sess) / (T * B)
    return T * B / (lat.mean() * 1e-3) / 1e9, np.percentile(lat, 50), np.percentile(lat, 99)

def two_independent(steps):
    lat = [[], []]
    def w(i):
        a = mine[i]
        for it in range(steps + 4):
            t0 = time.perf_counter()
            S[1 + i].lookup_packed(*a[it])
            if it >= 4: lat[i].append((time.perf_counter() - t0) * 1e3)
    mine = [[args(q, outs[i], ALL) for q in fresh(steps + 4)] for i in range(2)]
    th = [threading.Thread(target=w, args=(i,)) for i in range(2)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    l = np.array(lat[0] + lat[1])
    return 2 * (steps + 4) * T * B / dt / 1e9, np.percentile(l, 50), np.percentile(l, 99)

# parity of the split arrangement against the whole-request call
S[0].lookup_packed(*args(batches[3], outs[0], ALL)); torch.cuda.synchronize(); ref = outs[0].clone(); outs[0].zero_()
pool.run([(S[1 + i], args(batches[3], outs[0], p)) for i, p in enumerate(parts(2))]); torch.cuda.synchronize()
print("split-by-tables rows identical to the whole-request call:", bool(torch.equal(ref.view(torch.int32), outs[0].view(torch.int32))))
if os.environ.get("ONLY_TWO"):
    for rnd in range(3):
        g1, a50, a99 = one_in_flight(1, 40)
        g, p50, p99 = two_independent(60)
        print(f"[{os.environ.get('HPS_AMD_LIB_DIR', 'product lib')}] hit {HIT}: one in flight {g1:.3f} G p50 {a50:.3f} | two in flight {g:.3f} G lookups/s  p50 {p50:.3f} ms  p99 {p99:.3f} ms", flush=True)
    sys.exit(0)
for rnd in range(2):
    for k in (1, 2, 4):
        g, p50, p99 = one_in_flight(k, 40)
        print(f"one request in flight, {k} session(s): {g:.3f} G lookups/s  p50 {p50:.3f} ms  p99 {p99:.3f} ms  (hit rate of the last request {last_hit:.4f})", flush=True)
    if rnd == 1:
        for st_us in (100, 200, 300, 400):
            STAGGER_US = st_us
            g, p50, p99 = one_in_flight(2, 30)
            print(f"one request in flight, 2 sessions, second sub-call {st_us} us later: {g:.3f} G lookups/s  p50 {p50:.3f} ms  p99 {p99:.3f} ms", flush=True)
        STAGGER_US = 0
    g, p50, p99 = two_independent(40)
    print(f"two requests in flight, one session each: {g:.3f} G lookups/s  p50 {p50:.3f} ms  p99 {p99:.3f} ms  (hit rate of the last requests {1.0 - (S[1].last_stats().misses + S[2].last_stats().misses) / (2 * T * B):.4f})", flush=True)
