#!/usr/bin/env python3
"""Side benchmarks for the BASELINE configs that are not the headline (bench.py = config 2):

    python tests/tools/bench_configs.py c1     # 1 table 1,048,576 x 16, 4,096-key batch, CPU parameter server only
    python tests/tools/bench_configs.py c4     # two W&D models (D=[1,16], keys/sample [2,26], batch 1,024) sharing one GPU

Both go through the Triton plugin ABI (TRITONBACKEND_ModelInstanceExecute of libtriton_hps.so) driven by the mock
Triton core, i.e. what perf_analyzer would exercise against the reference (.gitlab-ci.yml:70).  Prints one JSON line.
"""
import json
import sys
import tempfile
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from tests import triton_mock as tm  # noqa: E402
from tests.conftest import ps_config  # noqa: E402
from oracle import hps_oracle as O  # noqa: E402

SEED = 20260929


def c1(iters=2000, warm=200):
    R, D, B = 1 << 20, 16, 4096
    tmp = Path(tempfile.mkdtemp())
    keys = np.arange(R, dtype=np.int64)
    rows = O.c_synth_rows(SEED, 0, 0, R, D)
    O.np_write_table(tmp / "t0", keys, rows)
    cfg = ps_config("c1", [(keys, rows)], dirs=[str(tmp / "t0")], gpucache=False, maxcat=[1], max_batch=B)
    (tmp / "ps.json").write_text(json.dumps(cfg))
    srv = tm.Server(tmp / "ps.json")
    inst = srv.load_model("c1", tm.model_config("c1", kind="KIND_CPU", gpus=[], max_batch_size=B)).create_instance("c1_0", tm.KIND_CPU)
    rng = np.random.default_rng(SEED)
    batches = [rng.integers(0, R, B).astype(np.int64) for _ in range(64)]
    nk = np.array([[B]], np.int32)
    lat = []
    ok = True
    co = O.COracle()
    co.add_table_arrays(keys, rows)
    for i in range(warm + iters):
        q = batches[i % len(batches)]
        req = tm.Request(str(i)).add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", nk).request_output()
        t0 = time.perf_counter()
        inst.execute([req])
        dt = time.perf_counter() - t0
        if i >= warm:
            lat.append(dt)
        if i % 500 == 0:
            ok &= req.error_code == -1 and np.array_equal(req.output_numpy().view(np.uint32), co.lookup(q, [B], [0.0]).view(np.uint32))
        req.close()
    srv.shutdown()
    lat = np.array(lat)
    # oracle (single thread) on the same batches, for scale
    t0 = time.perf_counter()
    for i in range(200):
        co.lookup(batches[i % len(batches)], [B], [0.0])
    cpu = 200 * B / (time.perf_counter() - t0)
    return {"config": "c1: 1 table 1,048,576 x 16 fp32, 4,096-key batch, CPU parameter server only (gpucache=false), one instance, via TRITONBACKEND_ModelInstanceExecute",
            "lookups_per_s": B / lat.mean(), "p50_ms": float(np.percentile(lat, 50) * 1e3), "p99_ms": float(np.percentile(lat, 99) * 1e3),
            "bit_exact_vs_oracle": bool(ok), "oracle_single_thread_lookups_per_s": cpu}


def c4(iters=300, warm=30, direct=False):
    import torch
    tmp = Path(tempfile.mkdtemp())
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8}, "models": []}
    tables = {}
    Rr = [1_000_000, 1_000_000]
    for m in ("wdl_a", "wdl_b"):
        tabs, dirs = [], []
        for t, D in enumerate([1, 16]):
            k = np.arange(Rr[t], dtype=np.int64)
            r = O.c_synth_rows(SEED + (m == "wdl_b"), t, 0, Rr[t], D)
            O.np_write_table(tmp / f"{m}_{t}", k, r)
            tabs.append((k, r))
            dirs.append(str(tmp / f"{m}_{t}"))
        tables[m] = tabs
        cfg["models"].append(ps_config(m, tabs, dirs=dirs, gpucache=True, gpucacheper=0.2, hit_rate_threshold=1.0, maxcat=[2, 26],
                                       max_batch=1024, extra={"ps_direct_access": bool(direct)})["models"][0])
    (tmp / "ps.json").write_text(json.dumps(cfg))
    srv = tm.Server(tmp / "ps.json")
    insts = {m: srv.load_model(m, tm.model_config(m, gpus=[0], max_batch_size=1024)).create_instance(f"{m}_0", tm.KIND_GPU, 0) for m in tables}
    B = 1024
    nk = [B * 2, B * 26]
    n_out = nk[0] + nk[1] * 16
    res = {}
    for hit in (0.5, 0.9, 0.99):
        lat = {m: [] for m in tables}
        bad = []

        def work(m, seed):
            rng = np.random.default_rng(seed)
            tabs = tables[m]
            out = torch.empty(n_out, dtype=torch.float32, device="cuda")
            for i in range(warm + iters):
                def draw(R, n):
                    hot = rng.random(n) < hit
                    return np.where(hot, rng.integers(0, int(0.2 * R) - 4096, n), rng.integers(int(0.2 * R), R, n))
                q = np.concatenate([draw(Rr[0], nk[0]), draw(Rr[1], nk[1])]).astype(np.int64)
                req = tm.Request(f"{m}{i}").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.asarray([nk], np.int32)).request_output()
                req.set_output_buffer(out.data_ptr(), n_out * 4, tm.MEM_GPU, 0)
                t0 = time.perf_counter()
                insts[m].execute([req])
                dt = time.perf_counter() - t0
                if i >= warm:
                    lat[m].append(dt)
                if i % 100 == 0:
                    ref = O.np_lookup(tabs, q, nk, [0.0, 0.0])
                    if req.error_code != -1 or not np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32)):
                        bad.append((m, i))
                req.close()
        th = [threading.Thread(target=work, args=(m, 7 + j)) for j, m in enumerate(tables)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        wall = time.perf_counter() - t0
        allv = np.concatenate([np.array(v) for v in lat.values()])
        res[f"target_hit_{hit}"] = {"requests_per_s_both_models": 2 * (warm + iters) / wall, "lookups_per_s": 2 * (warm + iters) * sum(nk) / wall,
                                    "p50_ms": float(np.percentile(allv, 50) * 1e3), "p99_ms": float(np.percentile(allv, 99) * 1e3),
                                    "bit_exact_vs_oracle": not bad}
    srv.shutdown()
    return {"config": "c4: two W&D models (tables 1M x [1,16], keys/sample [2,26], batch 1,024 = 28,672 keys/request), one instance each, "
                      "one MI355X, sync insert, host KEYS, device OUTPUT0, concurrent Execute via the mock Triton core", "results": res}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c1"
    if which == "c4":
        direct = len(sys.argv) > 2 and sys.argv[2] == "direct"
        out = c4(direct=direct)
        out["config"] += ", parameter-server tier: " + ("device-driven (ps_direct_access)" if direct else "host gather")
        print(json.dumps(out))
    else:
        print(json.dumps(c1()))
