"""Where does the time of a small request go?  One session, W&D-shaped model (D = [1, 16], 28,672 keys per request),
~99 % hit: wall time of the C-ABI call vs the GPU-side span (HIP events) vs the probe+gather kernel alone.

    python tools/small_request_breakdown.py [direct=1] [keys_on_device=1]
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from hugectr_backend_amd import hps
    direct = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
    on_dev = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    R, B = 1_000_000, 1024
    dims, maxcat = [1, 16], [2, 26]
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
           "models": [{"model": "m", "sparse_files": ["a", "b"], "num_of_worker_buffer_in_pool": 2,
                       "embedding_vecsize_per_table": dims, "maxnum_catfeature_query_per_table_per_sample": maxcat,
                       "default_value_for_each_table": [0.0, 0.0], "deployed_device_list": [0], "max_batch_size": B,
                       "gpucache": True, "gpucacheper": 0.2, "hit_rate_threshold": 1.0, "ps_direct_access": direct}]}
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t in range(2):
        ps.load_table_synthetic("m", t, 1, 0, R)
    ps.create_embedding_cache_per_model("m")
    s = hps.LookupSession.create(ps, "m", ps.get_embedding_cache("m", 0))
    s.set_option("timing", 1)
    rng = np.random.default_rng(0)
    nk = [B * 2, B * 26]
    out = torch.empty(nk[0] * 1 + nk[1] * 16, dtype=torch.float32, device="cuda")
    wall, gpu, ka, miss = [], [], [], []
    for it in range(400):
        q = np.concatenate([np.where(rng.random(n) < 0.99, rng.integers(0, int(0.2 * R) - 4096, n), rng.integers(int(0.2 * R), R, n))
                            for n in nk]).astype(np.int64)
        dq = torch.from_numpy(q).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if on_dev:
            s.lookup_device(dq, nk, out=out)
        else:
            s.lookup(q, nk, out=out)
        dt = (time.perf_counter() - t0) * 1e3
        st = s.last_stats()
        if it >= 50:
            wall.append(dt); gpu.append(st.gpu_call_ms); ka.append(st.probe_gather_ms); miss.append(st.misses)
    print(f"direct={int(direct)} keys_on_device={int(on_dev)}: {sum(nk)} keys/request, misses/request {np.mean(miss):.0f}: "
          f"wall p50 {np.percentile(wall, 50)*1e3:.0f} us, GPU-side span p50 {np.percentile(gpu, 50)*1e3:.0f} us, "
          f"probe+gather kernel p50 {np.percentile(ka, 50)*1e3:.0f} us")


if __name__ == "__main__":
    main()
