#!/usr/bin/env python3
"""Hit rate of the GPU cache over THOUSANDS of calls (bench.py's timed region is 260): the headline workload's key distribution on
a few tables, a fresh batch generated on the device for every call, keys in HBM, synchronous insertion.

    [HPS_LRU_INSERT_AGE=.. HPS_LRU_AGE_SHIFT=.. HPS_LRU_ADMIT=..] python tools/hit_rate_long.py --calls 4000 [--tables 4]

Prints the hit rate of every --report calls; tools/lru_sim.py is the CPU model of the same policy.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=4000)
    ap.add_argument("--report", type=int, default=500)
    ap.add_argument("--tables", type=int, default=4)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--cache-frac", type=float, default=0.2)
    ap.add_argument("--hit", type=float, default=0.957)
    ap.add_argument("--zipf", type=float, default=1.05)
    a = ap.parse_args()
    import torch
    from hugectr_backend_amd import build as hb, hps
    hb.build()
    T, R, D, Bn = a.tables, a.rows, a.dim, a.batch
    N = T * Bn
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
           "models": [{"model": "m", "sparse_files": [f"s{t}" for t in range(T)], "num_of_worker_buffer_in_pool": 2,
                       "embedding_vecsize_per_table": [D] * T, "maxnum_catfeature_query_per_table_per_sample": [1] * T,
                       "default_value_for_each_table": [0.0] * T, "deployed_device_list": [0], "max_batch_size": Bn,
                       "gpucache": True, "gpucacheper": a.cache_frac, "hit_rate_threshold": 1.0}]}
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t in range(T):
        ps.load_table_synthetic("m", t, B.SEED, 0, R)
    ps.create_embedding_cache_per_model("m")
    cache = ps.get_embedding_cache("m", 0)
    s = hps.LookupSession.create(ps, "m", cache)
    C = int(np.ceil(a.cache_frac * R))
    resident = []
    for t in range(T):
        k = np.arange(C, dtype=np.int64)
        resident.append(torch.from_numpy(k[cache.query(t, k) >= 0]).cuda())
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    cdf = torch.from_numpy(B.zipf_cdf(C, a.zipf)).cuda()
    out = torch.empty(N * D, dtype=torch.float32, device="cuda")
    nk = [Bn] * T
    misses = 0
    t0 = time.time()
    for call in range(1, a.calls + 1):
        b = B.make_batches_gpu(torch, gen, resident, cdf, R, C, Bn, a.hit, 1)[0]
        s.lookup_device(b, nk, out=out)
        misses += s.last_stats().misses
        if call % a.report == 0:
            print(f"call {call:5d}: hit rate of the last {a.report} calls {1.0 - misses / (a.report * N):.4f}  ({time.time() - t0:.0f} s)", flush=True)
            misses = 0
    print("counters", cache.counters())


if __name__ == "__main__":
    main()
