"""Host-tier gather rate at the shape of a config-2 miss batch: one 10 M x 128 table in RAM, requests of 82 K random keys
through hps_server_fetch (the serving pool fans them out), rows written into a reused buffer.  No GPU involved.

    python tools/host_gather_bench.py [rows=10000000] [keys=81920] [iters=60]
(The block size, prefetch hint and copy width this tool swept in rounds 2-3 are constants of csrc/ps/host_table.cpp since round 5.)
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    from hugectr_backend_amd import hps
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 81920
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    D = 128
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
           "models": [{"model": "m", "sparse_files": ["synthetic://0"], "num_of_worker_buffer_in_pool": 1,
                       "embedding_vecsize_per_table": [D], "maxnum_catfeature_query_per_table_per_sample": [1],
                       "default_value_for_each_table": [0.0], "deployed_device_list": [0], "max_batch_size": n,
                       "gpucache": False}]}
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    ps.load_table_synthetic("m", 0, 20260929, 0, R)
    rng = np.random.default_rng(0)
    qs = [rng.integers(0, R, n).astype(np.int64) for _ in range(8)]
    flat = np.zeros(n * D, np.float32)
    s = hps.LookupSession.create(ps, "m", None)   # the session's lookup goes through FetchMulti on the serving pool
    lat = []
    for i in range(iters + 10):
        q = qs[i % len(qs)]
        t0 = time.perf_counter()
        s.lookup(q, [n], out=flat)
        if i >= 10:
            lat.append(time.perf_counter() - t0)
    lat = np.array(lat)
    print(json.dumps({"rows": R, "keys_per_request": n, "p50_ms": float(np.median(lat) * 1e3),
                      "rows_per_s_M": n / float(np.median(lat)) / 1e6, "GB_per_s": n * D * 4 / float(np.median(lat)) / 1e9,
                      "env": {k: os.environ.get(k) for k in ("HPS_SERVING_THREADS",)}}))


if __name__ == "__main__":
    main()
