#!/usr/bin/env python3
"""Kernel A/B bench: probe+gather kernel variants on the Criteo-shaped workload, one session, interleaved rounds.

    python tools/kbench.py --variants 4,104,1004,1104,2,102,108 --rounds 5 [--rows 10000000] [--hit 1.0]

Variant code = U + 100*rolled_outer_loop + 1000*sampled_stamps (see LaunchProbeGather in csrc/cache/kernels.hip).
Prints median / min kernel time per variant (HIP events on the session's stream) and the implied fraction of the
8 TB/s HBM roofline at 1,032 algorithmic bytes per lookup.
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
    ap.add_argument("--variants", default="4,104,1004,1104")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--tables", type=int, default=26)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--cache-frac", type=float, default=0.2)
    ap.add_argument("--hit", type=float, default=1.1)
    ap.add_argument("--zipf", type=float, default=1.05)
    ap.add_argument("--balanced", default="1", help="comma list of 0/1: balanced grid (same chunk count per wave)")
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
    t0 = time.time()
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t in range(T):
        ps.load_table_synthetic("m", t, B.SEED, 0, R)
    ps.create_embedding_cache_per_model("m")
    cache = ps.get_embedding_cache("m", 0)
    s = hps.LookupSession.create(ps, "m", cache)
    s.set_option("timing", 1)
    C = int(np.ceil(a.cache_frac * R))
    resident = []
    for t in range(T):
        k = np.arange(C, dtype=np.int64)
        resident.append(torch.from_numpy(k[cache.query(t, k) >= 0]).cuda())
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    cdf = torch.from_numpy(B.zipf_cdf(C, a.zipf)).cuda()
    batches = B.make_batches_gpu(torch, gen, resident, cdf, R, C, Bn, a.hit, a.iters)
    out = torch.empty(N * D, dtype=torch.float32, device="cuda")
    nk = [Bn] * T
    print(f"setup {time.time() - t0:.1f}s", flush=True)
    variants = [(int(v), int(b)) for v in a.variants.split(",") for b in a.balanced.split(",")]
    res = {v: [] for v in variants}
    for v, bal in variants:  # warm
        s.set_option("probe_unroll", v)
        s.set_option("probe_balanced", bal)
        s.lookup_device(batches[0], nk, out=out)
    for r in range(a.rounds):
        for v, bal in variants:
            s.set_option("probe_unroll", v)
            s.set_option("probe_balanced", bal)
            for b in batches:
                s.lookup_device(b, nk, out=out)
                res[(v, bal)].append(s.last_stats().probe_gather_ms)
    alg = N * (8 + 8 * D)
    for v, bal in variants:
        x = np.array(res[(v, bal)])
        med, mn = float(np.median(x)), float(x.min())
        print(f"variant {v:5d} balanced={bal}: median {med:.4f} ms  min {mn:.4f} ms  frac(median) {alg / (med * 1e-3) / 8e12:.3f}  n={x.size}")


if __name__ == "__main__":
    main()
