"""TRITONBACKEND_ModelInstanceExecute of libtriton_hps.so on a real MI355X, driven by the mock Triton core:
GPU embedding cache enabled, OUTPUT0 in device memory (hps.cc:638-648).  Bit-exact against the CPU oracle."""
import json
import threading

import numpy as np
import pytest

from tests import triton_mock as tm
from tests.conftest import make_tables, ps_config

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _deploy(tmp_path, models, **ps_kw):
    """models: {name: (tables, maxcat, defaults)} -> (Server, ps_path)"""
    from oracle import hps_oracle as O
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8}, "models": []}
    for name, (tables, maxcat, defaults) in models.items():
        dirs = []
        for t, (k, r) in enumerate(tables):
            d = tmp_path / f"{name}_{t}"
            O.np_write_table(d, k, r)
            dirs.append(str(d))
        cfg["models"].append(ps_config(name, tables, dirs=dirs, gpucache=True, maxcat=maxcat, defaults=defaults, **ps_kw)["models"][0])
    ps_path = tmp_path / "ps.json"
    ps_path.write_text(json.dumps(cfg))
    return tm.Server(ps_path), ps_path


def _request(q, nk, out_elems, device_out=True, device_keys=False, rid="1"):
    import torch
    req = tm.Request(rid)
    if device_keys:
        dk = torch.from_numpy(q).cuda()
        req.add_input_raw("KEYS", tm.TYPE_INT64, [1, q.size], dk.data_ptr(), q.nbytes, tm.MEM_GPU, 0)
        req._keep.append(dk)
    else:
        req.add_input("KEYS", q.reshape(1, -1))
    req.add_input("NUMKEYS", np.asarray([nk], np.int32)).request_output("OUTPUT0")
    out = None
    if device_out:
        out = torch.full((max(out_elems, 1),), float("nan"), dtype=torch.float32, device="cuda")
        # the NaN fill runs on torch's stream and the backend writes on its own streams: without this the fill may
        # land after the lookup (Triton hands over buffers with no work pending on them)
        torch.cuda.synchronize()
        req.set_output_buffer(out.data_ptr(), out_elems * 4, tm.MEM_GPU, 0, keep=out)
    return req, out


def _result(req, out, n):
    import torch
    if out is not None:
        torch.cuda.synchronize()
        return out[:n].cpu().numpy()
    return req.output_numpy()


def test_wdl_gpu_cache_device_output_exact(tmp_path):
    from oracle import hps_oracle as O
    tables = make_tables([(3000, 1), (2000, 16)])
    srv, _ = _deploy(tmp_path, {"hps_wdl": (tables, [2, 26], [0.0, 0.0])}, gpucacheper=0.5, hit_rate_threshold=1.0)
    try:
        inst = srv.load_model("hps_wdl", tm.model_config("hps_wdl", gpus=[0])).create_instance("hps_wdl_0", tm.KIND_GPU, 0)
        rng = np.random.default_rng(0)
        for batch, device_out, device_keys in [(10, True, False), (10, True, True), (10, False, False), (1024, True, False), (1, True, True)]:
            nk = [batch * 2, batch * 26]
            q = np.concatenate([rng.choice(tables[0][0], nk[0]), rng.choice(tables[1][0], nk[1])]).astype(np.int64)
            q[::11] = -3 - np.arange(q[::11].size)  # absent keys -> default
            n = nk[0] * 1 + nk[1] * 16
            req, out = _request(q, nk, n, device_out, device_keys)
            inst.execute([req])
            assert (req.response_count, req.release_count, req.final, req.error_code) == (1, 1, True, -1), req.error_message
            name, dt, shape, ptr, nbytes, mt, mid = req.output(0)
            assert (name, dt, shape, nbytes) == ("OUTPUT0", tm.TYPE_FP32, [n], n * 4)
            assert mt == (tm.MEM_GPU if device_out else tm.MEM_CPU)
            assert req.int_param("NumSample") == batch and req.int_param("DeviceID") == 0
            ref = O.np_lookup(tables, q, nk, [0.0, 0.0])
            assert np.array_equal(_bits(_result(req, out, n)), _bits(ref)), (batch, device_out, device_keys)
    finally:
        srv.shutdown()


def test_an_instance_thread_joins_the_worker_pools_on_the_gpus_numa_node(tmp_path):
    """On a host with several NUMA nodes the plugin places the thread that executes an instance's requests on the node of the deployed GPU
    (where the worker pools and the tables are: csrc/ps/thread_pool.h) at its first request — unless somebody placed that thread inside one
    node already.  One-node host: nothing is touched.  Rows exact either way."""
    import glob
    import os
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    tables = make_tables([(5000, 8)])
    srv, _ = _deploy(tmp_path, {"numa": (tables, [4], [0.0])}, gpucacheper=0.5, hit_rate_threshold=1.0)
    got = {}

    def cpus(spec):
        out = set()
        for part in spec.split(","):
            a, _, b = part.partition("-")
            out.update(range(int(a), int(b or a) + 1))
        return out

    try:
        inst = srv.load_model("numa", tm.model_config("numa", gpus=[0])).create_instance("numa_0", tm.KIND_GPU, 0)
        rng = np.random.default_rng(4)

        def serve(tag, preset=None):
            if preset is not None:
                os.sched_setaffinity(0, preset)
            before = os.sched_getaffinity(0)
            nk = [400]
            q = rng.choice(tables[0][0], nk[0]).astype(np.int64)
            req, out = _request(q, nk, nk[0] * 8)
            inst.execute([req])
            assert req.error_code == -1, req.error_message
            assert np.array_equal(_bits(_result(req, out, nk[0] * 8)), _bits(O.np_lookup(tables, q, nk, [0.0])))
            got[tag] = (before, os.sched_getaffinity(0))

        node = hps.pool_numa_node()
        nodes = len(glob.glob("/sys/devices/system/node/node[0-9]*"))
        th = threading.Thread(target=serve, args=("free",))
        th.start(); th.join()
        before, after = got["free"]
        if nodes > 1 and node >= 0:
            want = cpus(open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()) & before
            assert after == want, (sorted(after)[:4], sorted(want)[:4])
            # a thread somebody confined to a few CPUs of the OTHER node keeps them
            # (the thread may set any CPU its cgroup allows, whatever mask it inherited)
            other = cpus(open(f"/sys/devices/system/node/node{1 - node if node < 2 else 0}/cpulist").read().strip())
            preset = set(sorted(other)[:3])
            try:
                os.sched_setaffinity(0, preset)
                os.sched_setaffinity(0, before | after)
            except OSError:
                preset = None      # a cpuset that does not include the other node: nothing to show
            if preset:
                th = threading.Thread(target=serve, args=("placed", preset))
                th.start(); th.join()
                assert got["placed"][1] == preset
        else:
            assert after == before
    finally:
        srv.shutdown()


@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_two_models_share_the_device_concurrently(tmp_path, direct):
    """BASELINE config 4 shape: two W&D models (D=[1,16], keys/sample [2,26], batch 1024), two instances each,
    mixed hit rate, concurrent Execute calls on one device."""
    from oracle import hps_oracle as O
    models = {m: (make_tables([(20000, 1), (20000, 16)], seed=s), [2, 26], [0.0, 0.0]) for m, s in (("wdl_a", 1), ("wdl_b", 2))}
    srv, _ = _deploy(tmp_path, models, gpucacheper=0.1, hit_rate_threshold=1.0, max_batch=1024,
                     extra={"ps_direct_access": direct})
    try:
        insts = {}
        for m in models:
            mod = srv.load_model(m, tm.model_config(m, gpus=[0], count=2))
            insts[m] = [mod.create_instance(f"{m}_{i}", tm.KIND_GPU, 0) for i in range(2)]
        errs = []

        def work(m, inst, seed):
            tables = models[m][0]
            rng = np.random.default_rng(seed)
            for it in range(10):
                batch = 1024
                nk = [batch * 2, batch * 26]
                # hot head (cached after first touch) + cold tail: hit rate between 50 and 99 %
                hot_frac = rng.uniform(0.5, 0.99)
                def draw(keys, n):
                    hot = rng.random(n) < hot_frac
                    return np.where(hot, rng.choice(keys[:1500], n), rng.choice(keys, n))
                q = np.concatenate([draw(tables[0][0], nk[0]), draw(tables[1][0], nk[1])]).astype(np.int64)
                n = nk[0] + nk[1] * 16
                req, out = _request(q, nk, n, True, it % 2 == 0, rid=f"{m}-{seed}-{it}")
                inst.execute([req])
                ref = O.np_lookup(tables, q, nk, [0.0, 0.0])
                if req.error_code != -1 or not np.array_equal(_bits(_result(req, out, n)), _bits(ref)):
                    errs.append((m, seed, it, req.error_message))
        th = [threading.Thread(target=work, args=(m, inst, 10 * j + i)) for j, m in enumerate(models) for i, inst in enumerate(insts[m])]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs[:3]
    finally:
        srv.shutdown()


@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_requests_of_one_execute_call_are_served_by_one_engine_call(tmp_path, direct):
    """TRITONBACKEND_ModelInstanceExecute with 1 / 3 / 8 requests (Triton's dynamic batcher hands small requests over together; the
    reference runs one blocking lookup per request, hps.cc:406): the valid requests go as ONE engine call — one probe / gather /
    miss path for all their keys — and every request's rows land in its own output buffer, device or host, in its own table-major
    order; an invalid request in the middle gets its error response and takes no part; requests whose KEYS are already on the
    device, or that are too big together, are served one by one as before.  Bit-exact, one final response and one release each."""
    import torch
    from oracle import hps_oracle as O
    tables = make_tables([(30000, 1), (20000, 16), (5000, 7)], seed=5)
    defaults = [0.0, 0.25, -1.0]
    srv, _ = _deploy(tmp_path, {"mr": (tables, [2, 26, 3], defaults)}, gpucacheper=0.2, hit_rate_threshold=1.0, max_batch=1024,
                     extra={"ps_direct_access": direct})
    try:
        inst = srv.load_model("mr", tm.model_config("mr", gpus=[0], max_batch_size=1024)).create_instance("mr_0", tm.KIND_GPU, 0)
        rng = np.random.default_rng(17 + direct)

        def make(batch, device_out, device_keys=False, rid="r"):
            nk = [batch * 2, batch * 26, batch * 3]
            q = np.concatenate([rng.choice(k, n) for (k, _), n in zip(tables, nk)]).astype(np.int64)
            q[::9] = -7 - rng.integers(0, 1 << 30, q[::9].size)          # keys that exist nowhere -> the table's default
            n = nk[0] * 1 + nk[1] * 16 + nk[2] * 7
            req, out = _request(q, nk, n, device_out, device_keys, rid=rid)
            return req, out, n, O.np_lookup(tables, q, nk, defaults), batch

        for count in (1, 3, 8):
            for trial in range(3):
                items = [make(int(rng.integers(1, 1024 // count + 1)), bool((i + trial) % 3), rid=f"{count}.{trial}.{i}") for i in range(count)]
                bad = None
                if count >= 3:
                    # NUMKEYS with one table too few, in the middle of the call
                    qb = np.zeros(10, np.int64)
                    bad = tm.Request("bad").add_input("KEYS", qb.reshape(1, -1)).add_input("NUMKEYS", np.asarray([[4, 6]], np.int32)).request_output()
                reqs = [it[0] for it in items]
                if bad is not None:
                    reqs.insert(count // 2, bad)
                before = inst.stats()
                inst.execute(reqs)
                for req, out, n, ref, batch in items:
                    assert (req.response_count, req.release_count, req.final, req.error_code) == (1, 1, True, -1), (req.id, req.error_message)
                    assert np.array_equal(_bits(_result(req, out, n)), _bits(ref)), (count, trial, req.id)
                    assert req.int_param("NumSample") == batch and req.int_param("DeviceID") == 0
                    assert req.output(0)[5] == (tm.MEM_GPU if out is not None else tm.MEM_CPU)
                st = inst.stats()
                assert st.last_batch_size == sum(it[4] for it in items)
                assert st.last_distinct_compute_starts == 1, (count, st.last_distinct_compute_starts)      # ONE lookup
                if bad is not None:
                    assert (bad.response_count, bad.release_count, bad.final) == (1, 1, True) and bad.error_code == tm.ERR["INVALID_ARG"]
                    assert st.failed_requests - before.failed_requests == 1
        # KEYS already in device memory: such requests are served one by one (no host copy of the keys exists to merge)
        items = [make(20, True, device_keys=(i == 1), rid=f"dk{i}") for i in range(3)]
        inst.execute([it[0] for it in items])
        for req, out, n, ref, _ in items:
            assert req.error_code == -1 and np.array_equal(_bits(_result(req, out, n)), _bits(ref)), req.id
        assert inst.stats().last_distinct_compute_starts == 3
        # together more samples than one call holds (max_batch_size): one by one
        items = [make(700, True, rid=f"big{i}") for i in range(2)]
        inst.execute([it[0] for it in items])
        for req, out, n, ref, _ in items:
            assert req.error_code == -1 and np.array_equal(_bits(_result(req, out, n)), _bits(ref)), req.id
        assert inst.stats().last_distinct_compute_starts == 2
        torch.cuda.synchronize()
    finally:
        srv.shutdown()


@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_new_model_version_refreshes_the_cache(tmp_path, direct):
    """Loading version 2 of a served model re-reads the sparse files and refreshes the device cache
    asynchronously (hps.cc:207-226, model_state.cpp:124-142,413-418)."""
    import time
    from oracle import hps_oracle as O
    tables = make_tables([(1000, 16)])
    keys, rows = tables[0]
    srv, ps_path = _deploy(tmp_path, {"m": (tables, [1], [0.0])}, gpucacheper=1.0, hit_rate_threshold=1.0, max_batch=2048,
                           extra={"ps_direct_access": direct})
    try:
        m1 = srv.load_model("m", tm.model_config("m", gpus=[0]), version=1)
        i1 = m1.create_instance("m_v1", tm.KIND_GPU, 0)
        q = keys[:256].astype(np.int64)
        req, out = _request(q, [256], 256 * 16)
        i1.execute([req])
        assert np.array_equal(_bits(_result(req, out, 256 * 16)), _bits(rows[:256].ravel()))
        # retrained model: same keys, new vectors written over the sparse files, deployed as version 2
        rows2 = (rows * 0.5 + 0.125).astype(np.float32)
        O.np_write_table(tmp_path / "m_0", keys, rows2)
        m2 = srv.load_model("m", tm.model_config("m", gpus=[0]), version=2)
        i2 = m2.create_instance("m_v2", tm.KIND_GPU, 0)
        deadline = time.time() + 30
        ok = False
        while time.time() < deadline and not ok:
            req, out = _request(q, [256], 256 * 16)
            i2.execute([req])
            ok = np.array_equal(_bits(_result(req, out, 256 * 16)), _bits(rows2[:256].ravel()))
            if not ok:
                time.sleep(0.2)
        assert ok, "cache was not refreshed with the new version's vectors"
    finally:
        srv.shutdown()


def test_instance_on_undeployed_device_is_rejected(tmp_path):
    tables = make_tables([(100, 4)])
    srv, _ = _deploy(tmp_path, {"m": (tables, [1], [0.0])})
    try:
        with pytest.raises(tm.TritonError) as e:   # model_state.cpp:396-402
            srv.load_model("m", tm.model_config("m", gpus=[3]))
        assert e.value.code == tm.ERR["INVALID_ARG"] and "deployed_device_list" in e.value.msg
        with pytest.raises(tm.TritonError) as e:   # KIND_CPU instance for a GPU-cache model (model_state.cpp:287-290)
            srv.load_model("m", tm.model_config("m", kind="KIND_CPU", gpus=[]))
        assert "must be KIND_GPU" in e.value.msg
    finally:
        srv.shutdown()


def test_large_request_keys_staged_narrowed_pinned_or_split_over_buffers(tmp_path):
    """A request large enough for the key staging to go through the serving pool (>= 131,072 keys) through the plugin:
    pageable KEYS that fit 32 bits (cross PCIe as uint32), the same with one 41-bit key (8-byte fallback), KEYS in
    page-locked memory reported as CPU_PINNED (DMA in place), and KEYS delivered in two buffers (concatenated by the
    shell) — identical rows every time."""
    import torch
    from oracle import hps_oracle as O
    tables = make_tables([(80000, 8), (60000, 4)], seed=21)
    wide = tables[1][0].copy()
    wide[:50] += 1 << 40
    tables[1] = (wide, tables[1][1])
    srv, _ = _deploy(tmp_path, {"big": (tables, [1, 1], [0.5, -1.0])}, gpucacheper=0.4, hit_rate_threshold=1.0, max_batch=200000)
    try:
        inst = srv.load_model("big", tm.model_config("big", gpus=[0], max_batch_size=200000)).create_instance("big_0", tm.KIND_GPU, 0)
        rng = np.random.default_rng(5)
        nk = [150000, 120000]
        q = np.concatenate([rng.choice(tables[0][0], nk[0]), rng.choice(wide[50:], nk[1])]).astype(np.int64)
        q[::101] = (1 << 31) + np.arange(q[::101].size)      # absent, 32-bit
        n_out = nk[0] * 8 + nk[1] * 4

        def run(keys, mode):
            out = torch.full((n_out,), float("nan"), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            req = tm.Request(mode)
            keep = None
            if mode == "pinned":
                keep = torch.from_numpy(keys).pin_memory()
                req.add_input_raw("KEYS", tm.TYPE_INT64, [1, keys.size], keep.data_ptr(), keys.nbytes, tm.MEM_CPU_PINNED, 0)
            elif mode == "two_buffers":
                a, b = np.ascontiguousarray(keys[:100001]), np.ascontiguousarray(keys[100001:])
                req.add_input_raw("KEYS", tm.TYPE_INT64, [1, keys.size], a.ctypes.data, a.nbytes, tm.MEM_CPU, 0)
                # the mock appends a second buffer to the same tensor
                from ctypes import c_void_p
                shp = np.asarray([1, keys.size], dtype=np.int64)
                tm._check(tm.lib().mock_request_add_input_buffer(req._h, b"KEYS", tm.TYPE_INT64, shp.ctypes.data, 2, c_void_p(b.ctypes.data),
                                                                 b.nbytes, tm.MEM_CPU, 0))
                keep = (a, b, shp)
            else:
                req.add_input("KEYS", keys.reshape(1, -1))
            req.add_input("NUMKEYS", np.asarray([nk], np.int32)).request_output("OUTPUT0")
            req.set_output_buffer(out.data_ptr(), n_out * 4, tm.MEM_GPU, 0, keep=out)
            inst.execute([req])
            assert (req.response_count, req.release_count, req.error_code) == (1, 1, -1), (mode, req.error_message)
            torch.cuda.synchronize()
            res = out.cpu().numpy()
            req.close()
            del keep
            return res

        ref = O.np_lookup(tables, q, nk, [0.5, -1.0])
        for mode in ("pageable", "pageable", "pinned", "two_buffers"):
            assert np.array_equal(_bits(run(q, mode)), _bits(ref)), mode
        q2 = q.copy()
        q2[nk[0] + 7] = wide[3]          # a 41-bit key: this request cannot be narrowed
        ref2 = O.np_lookup(tables, q2, nk, [0.5, -1.0])
        assert np.array_equal(_bits(run(q2, "pageable")), _bits(ref2))
        assert np.array_equal(_bits(run(q, "pageable")), _bits(ref))
    finally:
        srv.shutdown()
