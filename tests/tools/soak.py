"""Soak test on the GPU box: several sessions hammer one cache with random batches while another thread keeps
refreshing the cache and reloading the host tables (same content), for a fixed time.  Every answer is checked bit for bit
(tables use keys 0..R-1, so the expected row of key k is rows[k]; absent keys must come back as the default).

    python tests/tools/soak.py [seconds=60] [direct=1] [sessions=3] [threshold=1.0]
"""
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))


def main():
    import torch
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    direct = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    nsess = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    thr = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    T, R, B = 8, 400_000, 16384
    dims = [128, 64, 32, 16, 128, 8, 4, 1]
    defaults = [0.25 * (t + 1) for t in range(T)]
    tables = [(np.arange(R, dtype=np.int64), O.c_synth_rows(O.SEED, t, 0, R, dims[t])) for t in range(T)]
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
           "models": [{"model": "soak", "sparse_files": [f"x{t}" for t in range(T)], "num_of_worker_buffer_in_pool": 3,
                       "embedding_vecsize_per_table": dims, "maxnum_catfeature_query_per_table_per_sample": [1] * T,
                       "default_value_for_each_table": defaults, "deployed_device_list": [0], "max_batch_size": B,
                       "gpucache": True, "gpucacheper": 0.05, "hit_rate_threshold": thr, "ps_direct_access": direct}]}
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays("soak", t, k, r)
    ps.create_embedding_cache_per_model("soak")
    cache = ps.get_embedding_cache("soak", 0)
    sessions = [hps.LookupSession.create(ps, "soak", cache) for _ in range(nsess)]
    stop = threading.Event()
    errs, calls = [], [0] * nsess

    def worker(si):
        rng = np.random.default_rng(1000 + si)
        s = sessions[si]
        try:
            while not stop.is_set():
                nk = [int(rng.integers(0, B + 1)) for _ in range(T)]
                parts = []
                for t in range(T):
                    hot = rng.random(nk[t]) < 0.8
                    qq = np.where(hot, rng.integers(0, R // 10, nk[t]), rng.integers(0, R, nk[t]))
                    qq = np.where(rng.random(nk[t]) < 0.01, R + rng.integers(0, 1 << 40, nk[t]), qq)   # 1 % absent
                    parts.append(qq.astype(np.int64))
                q = np.concatenate(parts)
                if si % 2:
                    out = s.lookup_device(torch.from_numpy(q).cuda(), nk).cpu().numpy()
                else:
                    out = s.lookup(q, nk).cpu().numpy()
                off = oo = 0
                for t in range(T):
                    qq = q[off:off + nk[t]]
                    got = out[oo:oo + nk[t] * dims[t]].reshape(nk[t], dims[t])
                    present = qq < R
                    exp = np.full((nk[t], dims[t]), np.float32(defaults[t]), np.float32)
                    exp[present] = tables[t][1][qq[present]]
                    same = got.view(np.uint32) == exp.view(np.uint32)
                    if thr < 1.0:     # async tables may answer the default for keys that are not resident yet
                        same |= (got == np.float32(defaults[t]))
                    if not same.all():
                        errs.append((si, t, int((~same).sum())))
                        stop.set()
                        return
                    off += nk[t]
                    oo += nk[t] * dims[t]
                calls[si] += 1
        except Exception as e:  # noqa: BLE001
            errs.append((si, repr(e)))
            stop.set()

    refreshed = [0, 0]

    def churn():
        i = 0
        try:
            while not stop.is_set():
                time.sleep(0.05)
                if i % 3 == 0:
                    t = (i // 3) % T
                    ps.load_table_arrays("soak", t, *tables[t])       # reload (same content) under the sessions
                # the default refresh (only what can differ: here the reloaded table), every fifth time the reference's full pass
                st = ps.refresh_embedding_cache("soak", 0, full=(i % 5 == 4))
                refreshed[0] += st["rows_refreshed"]
                refreshed[1] += st["tables_unchanged"]
                i += 1
        except Exception as e:  # noqa: BLE001
            errs.append(("churn", repr(e)))
            stop.set()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(nsess)] + [threading.Thread(target=churn)]
    t0 = time.time()
    [x.start() for x in th]
    while time.time() - t0 < seconds and not stop.is_set():
        time.sleep(0.5)
    stop.set()
    [x.join() for x in th]
    cache.wait_async()
    print(f"soak {'direct' if direct else 'host'} thr={thr}: {sum(calls)} lookups by {nsess} sessions in {time.time()-t0:.0f} s, "
          f"errors: {errs if errs else 'none'}, counters {cache.counters()}, refresh: {refreshed[0]} rows re-read, {refreshed[1]} table passes skipped as unchanged, "
          f"fork-join overruns {hps.pool_fast_overruns()}")
    sys.exit(1 if errs or hps.pool_fast_overruns() else 0)


if __name__ == "__main__":
    main()
