import sys, threading, numpy as np
sys.path.insert(0, '.')
from tests.conftest import make_tables, ps_config
from tests.test_gpu_lookup import _mk, _queries, _bits
from hugectr_backend_amd import hps
from oracle import hps_oracle as O

def run(mode, frac):
    tables = make_tables([(6000, 128), (6000, 16)])
    ps, cache, s0 = _mk("shared"+mode+str(frac), tables, maxcat=[1, 1], gpucacheper=frac, max_batch=4096)
    s1 = hps.LookupSession.create(ps, ps_name:=("shared"+mode+str(frac)), cache)
    co = O.COracle()
    for k, r in tables: co.add_table_arrays(k, r)
    glock = threading.Lock()
    res = {0: [], 1: []}
    def worker(sess, sid, seed):
        rng = np.random.default_rng(seed)
        for it in range(8):
            nk = [4096, 4096]
            q = _queries(rng, tables, nk, miss_frac=0.05)
            if mode == "locked":
                with glock: out = sess.lookup(q, nk).cpu().numpy()
            else:
                out = sess.lookup(q, nk).cpu().numpy()
            ref = co.lookup(q, nk, [0.0, 0.0])
            bad0 = (_bits(out[:4096*128]).reshape(4096,128) != _bits(ref[:4096*128]).reshape(4096,128)).any(axis=1)
            bad1 = (_bits(out[4096*128:]).reshape(4096,16) != _bits(ref[4096*128:]).reshape(4096,16)).any(axis=1)
            res[sid].append((int(bad0.sum()), int(bad1.sum())))
    if mode == "seq":
        for it in range(4):
            worker(s0, 0, 100+it); worker(s1, 1, 200+it)
    else:
        th = [threading.Thread(target=worker, args=(s0, 0, 100)), threading.Thread(target=worker, args=(s1, 1, 200))]
        [t.start() for t in th]; [t.join() for t in th]
    print(mode, frac, "s0", res[0][:8], "s1", res[1][:8], flush=True)

for mode in ["seq", "locked", "threads"]:
    for frac in [0.05, 1.0]:
        run(mode, frac)
