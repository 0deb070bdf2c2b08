#!/usr/bin/env python3
"""Kernel A/B bench: probe-kernel variants and gather walks on the Criteo-shaped workload, one session, interleaved rounds.

    python tools/kbench.py --variants 2,4,8,102,104,108 --xcd 0,1 --rounds 5 [--rows 10000000] [--hit 1.0]

Variant code = U + 100*no_dedup (see LaunchProbeTiles in csrc/cache/kernels.hip); --xcd: chunk walk of the gather kernel
(1 = each XCD sweeps its own eighth of the key range).  Prints median / min of the probe kernels (K_P + K_M) and of
the gather kernel K_G (HIP events on the session's stream), and the pair's fraction of the 8 TB/s HBM roofline at
SURVEY 8(d)'s 1,032 algorithmic bytes per lookup.
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
    ap.add_argument("--variants", default="1002,1102")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--tables", type=int, default=26)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--cache-frac", type=float, default=0.2)
    ap.add_argument("--hit", type=float, default=1.1)
    ap.add_argument("--zipf", type=float, default=1.05)
    ap.add_argument("--xcd", default="1", help="comma list of 0/1: XCD-aware chunk walk of the gather kernel")
    ap.add_argument("--threshold-permille", type=int, default=1000, help="<1000: the policy needs the unique-key count (K_H: distinct slots in LDS bitmaps)")
    ap.add_argument("--cache-type", default="dynamic", help="embedding_cache_type: static = no recency stamps written, no inserts")
    ap.add_argument("--env", default="", help="NAME=v1,v2: an environment switch the engine reads per launch, varied like the variants")
    a = ap.parse_args()
    import os
    env_name, env_vals = (a.env.split("=")[0], a.env.split("=")[1].split(",")) if a.env else ("", [""])
    import torch
    from hugectr_backend_amd import build as hb, hps
    hb.build()
    T, R, D, Bn = a.tables, a.rows, a.dim, a.batch
    N = T * Bn
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
           "models": [{"model": "m", "sparse_files": [f"s{t}" for t in range(T)], "num_of_worker_buffer_in_pool": 2,
                       "embedding_vecsize_per_table": [D] * T, "maxnum_catfeature_query_per_table_per_sample": [1] * T,
                       "default_value_for_each_table": [0.0] * T, "deployed_device_list": [0], "max_batch_size": Bn,
                       "gpucache": True, "gpucacheper": a.cache_frac, "hit_rate_threshold": 1.0, "embedding_cache_type": a.cache_type}]}
    t0 = time.time()
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t in range(T):
        ps.load_table_synthetic("m", t, B.SEED, 0, R)
    ps.create_embedding_cache_per_model("m")
    cache = ps.get_embedding_cache("m", 0)
    s = hps.LookupSession.create(ps, "m", cache)
    s.set_option("timing", 1)
    s.set_option("hit_rate_threshold_permille", a.threshold_permille)
    C = int(np.ceil(a.cache_frac * R))
    resident = []
    for t in range(T):
        k = np.arange(C, dtype=np.int64)
        resident.append(torch.from_numpy(k[cache.query(t, k) >= 0]).cuda())
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    cdf = torch.from_numpy(B.zipf_cdf(C, a.zipf)).cuda()
    variants = [(int(v), int(b), ev) for v in a.variants.split(",") for b in a.xcd.split(",") for ev in env_vals]
    # with misses in the batches every batch is used ONCE (a replayed batch finds its cold keys inserted): one batch per
    # variant and step, the variants taking turns
    fresh = a.hit < 1.0
    batches = B.make_batches_gpu(torch, gen, resident, cdf, R, C, Bn, a.hit, a.iters * (a.rounds * len(variants) + 1) if fresh else a.iters)
    out = torch.empty(N * D, dtype=torch.float32, device="cuda")
    nk = [Bn] * T
    print(f"setup {time.time() - t0:.1f}s", flush=True)
    res = {v: [] for v in variants}
    for i, (v, x, ev) in enumerate(variants):  # warm
        if env_name: os.environ[env_name] = ev
        s.set_option("probe_variant", v)
        s.set_option("xcd_walk", x)
        s.lookup_device(batches[i % a.iters], nk, out=out)
    nxt = a.iters   # (fresh: the first a.iters batches warmed the variants up)
    for r in range(a.rounds):
        for v, x, ev in variants:
            if env_name: os.environ[env_name] = ev
            s.set_option("probe_variant", v)
            s.set_option("xcd_walk", x)
            for b in (batches[nxt:nxt + a.iters] if fresh else batches):
                s.lookup_device(b, nk, out=out)
                st = s.last_stats()
                res[(v, x, ev)].append((st.probe_gather_ms, st.hit_gather_ms, st.scatter_ms, st.insert_ms))
            nxt += a.iters if fresh else 0
    alg = N * (8 + 8 * D)
    for v, x, ev in variants:
        arr = np.array(res[(v, x, ev)])
        p, g = float(np.median(arr[:, 0])), float(np.median(arr[:, 1]))
        print(f"variant {v:4d} xcd_walk={x} {env_name}={ev}: probe median {p * 1e3:7.1f} us (min {arr[:, 0].min() * 1e3:7.1f})  gather median {g * 1e3:7.1f} us "
              f"(min {arr[:, 1].min() * 1e3:7.1f})  scatter median {np.median(arr[:, 2]) * 1e3:6.1f}  insert median {np.median(arr[:, 3]) * 1e3:6.1f} us  "
              f"frac(probe+gather) {alg / ((p + g) * 1e-3) / 8e12:.3f}  n={arr.shape[0]}")


if __name__ == "__main__":
    main()
