"""The seven TRITONBACKEND_* entry points of libtriton_hps.so driven by the mock Triton core, CPU-only models
(ps.json "gpucache": false -> rows straight from the host parameter server into a TRITONSERVER_MEMORY_CPU
output: /root/reference/hps_backend/src/hps.cc:638-642,686-690; test/triton_server.sh:45-52 `ps_cpu.json`).

Reads like a perf_analyzer / python-client session against the reference: start server with
--backend-config=hps,ps=<file>, load model from its config, send KEYS/NUMKEYS, read OUTPUT0.
"""
import json
import subprocess

import numpy as np
import pytest

from tests import triton_mock as tm
from tests.conftest import make_tables, ps_config


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture()
def wdl_server(tmp_path):
    """W&D sample deployment: 2 tables D=[1,16], keys/sample [2,26], max_batch 1024 (README.md:143-156)."""
    from oracle import hps_oracle as O
    tables = make_tables([(3000, 1), (2000, 16)])
    dirs = []
    for t, (k, r) in enumerate(tables):
        d = tmp_path / f"wdl{t}_sparse_2000.model"
        O.np_write_table(d, k, r)
        dirs.append(str(d))
    cfg = ps_config("hps_wdl", tables, dirs=dirs, gpucache=False, maxcat=[2, 26], defaults=[0.0, 0.0], max_batch=1024)
    ps_path = tmp_path / "ps_cpu.json"
    ps_path.write_text(json.dumps(cfg))
    srv = tm.Server(ps_path)
    yield srv, tables, ps_path, tmp_path
    srv.shutdown()


def _wdl_request(rng, tables, batch=10, rid="1"):
    nk = np.array([[batch * 2, batch * 26]], dtype=np.int32)            # row_ptrs of the sample client
    q = np.concatenate([rng.choice(tables[0][0], batch * 2), rng.choice(tables[1][0], batch * 26)]).astype(np.int64)
    req = tm.Request(rid).add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", nk).request_output("OUTPUT0")
    return req, q, nk.ravel()


def test_library_exports_exactly_the_seven_entry_points():
    out = subprocess.run(["nm", "-D", "--defined-only", str(tm.BACKEND_LIB)], capture_output=True, text=True, check=True).stdout
    syms = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert syms == sorted(tm.EXPORTS)  # libtriton_hps.ldscript:26-30


def test_wdl_request_matches_reference_sample_shape_and_oracle(wdl_server):
    """10 samples -> OUTPUT0 shape [4180], parameters NumSample=10, DeviceID=0
    (samples/Hierarchical_Parameter_Server_Deployment.ipynb:738-747,793-795)."""
    from oracle import hps_oracle as O
    srv, tables, _, _ = wdl_server
    model = srv.load_model("hps_wdl", tm.model_config("hps_wdl", kind="KIND_CPU", gpus=[]))
    inst = model.create_instance("hps_wdl_0", tm.KIND_CPU, 0)
    req, q, nk = _wdl_request(np.random.default_rng(0), tables)
    inst.execute([req])
    assert (req.response_count, req.release_count, req.final, req.error_code) == (1, 1, True, -1)
    name, dt, shape, ptr, nbytes, mt, _ = req.output(0)
    assert (name, dt, shape, nbytes, mt) == ("OUTPUT0", tm.TYPE_FP32, [4180], 4180 * 4, tm.MEM_CPU)
    assert req.int_param("NumSample") == 10 and req.int_param("DeviceID") == 0
    ref = O.np_lookup(tables, q, nk, [0.0, 0.0])
    assert np.array_equal(_bits(req.output_numpy()), _bits(ref))
    st = inst.stats()
    assert (st.success_requests, st.failed_requests, st.batch_reports) == (1, 0, 1)
    assert st.last_batch_size == 10


def test_several_requests_in_one_execute_call_and_missing_keys_get_default(tmp_path):
    from oracle import hps_oracle as O
    tables = make_tables([(500, 16)])
    d = tmp_path / "t0"
    O.np_write_table(d, *tables[0])
    cfg = ps_config("m", tables, dirs=[str(d)], gpucache=False, maxcat=[3], defaults=[1.0], max_batch=1024)
    (tmp_path / "ps.json").write_text(json.dumps(cfg))
    srv = tm.Server(tmp_path / "ps.json")
    try:
        inst = srv.load_model("m", tm.model_config("m", kind="KIND_CPU", gpus=[])).create_instance("m_0", tm.KIND_CPU)
        rng = np.random.default_rng(1)
        reqs, refs = [], []
        for i in range(4):
            n = 3 * (i + 1) * 16   # e.g. NUMKEYS [[3072]] style single-table request (02_...ipynb:661-662)
            q = rng.choice(tables[0][0], n).astype(np.int64)
            q[::7] = -5 - np.arange(q[::7].size)  # keys that exist nowhere -> default_value_for_each_table = 1.0
            reqs.append(tm.Request(str(i)).add_input("KEYS", q.reshape(1, -1))
                        .add_input("NUMKEYS", np.array([[n]], np.int32)).request_output())
            refs.append(O.np_lookup(tables, q, [n], [1.0]))
        inst.execute(reqs)
        for r, ref in zip(reqs, refs):
            assert (r.response_count, r.release_count, r.error_code) == (1, 1, -1)
            out = r.output_numpy()
            assert np.array_equal(_bits(out), _bits(ref))
            assert (out.reshape(-1, 16)[0] == 1.0).all()
        assert inst.stats().success_requests == 4
        # the four small requests were served by ONE lookup (the reference: one blocking lookup per request, hps.cc:406)
        assert inst.stats().last_distinct_compute_starts == 1
    finally:
        srv.shutdown()


def test_requests_of_one_execute_call_share_one_lookup_and_keep_their_own_verdicts(wdl_server):
    """Several requests in one TRITONBACKEND_ModelInstanceExecute call (Triton's dynamic batcher): the valid ones are served by
    ONE engine call — keys concatenated table by table, every request's rows delivered to its own OUTPUT0 in its own order —, a
    request that fails validation in the middle gets its error response and takes no part, a request that wants no output gets its
    empty response; exactly one final response and one release each.  Two tables of different widths, mixed sizes, an empty
    request, keys split over several buffers."""
    from oracle import hps_oracle as O
    srv, tables, _, _ = wdl_server
    inst = srv.load_model("hps_wdl", tm.model_config("hps_wdl", kind="KIND_CPU", gpus=[])).create_instance("i", tm.KIND_CPU)
    rng = np.random.default_rng(11)
    for count in (1, 3, 8):
        reqs, want = [], []
        for i in range(count):
            batch = int(rng.integers(1, 40))
            r, q, nk = _wdl_request(rng, tables, batch=batch, rid=f"{count}-{i}")
            reqs.append(r)
            want.append(O.np_lookup(tables, q, nk, [0.0, 0.0]))
        extra = []
        if count >= 3:
            # in the middle: NUMKEYS that does not add up; a request that wants no output; an empty request; keys in three buffers
            _, q, nk = _wdl_request(rng, tables, batch=4, rid="x")
            bad = tm.Request("bad").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.array([[nk[0], nk[1] - 1]], np.int32)).request_output()
            silent = tm.Request("silent").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.array([nk], np.int32))
            empty = tm.Request("empty").add_input("KEYS", np.zeros((1, 0), np.int64)).add_input("NUMKEYS", np.array([[0, 0]], np.int32)).request_output()
            split = tm.Request("split")
            pieces = [q[:3].copy(), q[3:50].copy(), q[50:].copy()]
            for piece in pieces:
                split.add_input_raw("KEYS", tm.TYPE_INT64, [1, q.size], piece.ctypes.data, piece.nbytes, tm.MEM_CPU)
            split._keep += pieces
            split.add_input("NUMKEYS", np.array([nk], np.int32)).request_output()
            reqs[1:1] = [bad, silent]
            want[1:1] = [None, None]
            reqs += [empty, split]
            want += [np.zeros(0, np.float32), O.np_lookup(tables, q, nk, [0.0, 0.0])]
            extra = [bad, silent]
        before = inst.stats()
        inst.execute(reqs)
        for r, ref in zip(reqs, want):
            assert (r.response_count, r.release_count, r.final) == (1, 1, True), r.id
            if r in extra:
                continue
            assert r.error_code == -1, (r.id, r.error_message)
            assert np.array_equal(_bits(r.output_numpy()), _bits(ref)), (count, r.id)
        st = inst.stats()
        if extra:
            assert extra[0].error_code == tm.ERR["INVALID_ARG"] and extra[1].error_code == -1 and extra[1].output_count == 0
        assert st.failed_requests - before.failed_requests == (1 if extra else 0)
        assert st.success_requests - before.success_requests == len(reqs) - (1 if extra else 0)
        # one lookup for all requests that had rows to fetch (the silent one is answered without one and keeps its own start time)
        assert st.last_distinct_compute_starts == (1 if count == 1 else 2 if extra else 1), (count, st.last_distinct_compute_starts)


def test_keys_delivered_in_several_buffers_are_concatenated(wdl_server):
    """The reference overwrites offset 0 for every buffer (hps.cc:586-597); here the pieces are concatenated."""
    from oracle import hps_oracle as O
    srv, tables, _, _ = wdl_server
    inst = srv.load_model("hps_wdl", tm.model_config("hps_wdl", kind="KIND_CPU", gpus=[])).create_instance("i", tm.KIND_CPU)
    rng = np.random.default_rng(2)
    batch = 8
    q = np.concatenate([rng.choice(tables[0][0], batch * 2), rng.choice(tables[1][0], batch * 26)]).astype(np.int64)
    a, b, c = q[:5].copy(), q[5:100].copy(), q[100:].copy()
    req = tm.Request("split")
    for piece in (a, b, c):
        req.add_input_raw("KEYS", tm.TYPE_INT64, [1, q.size], piece.ctypes.data, piece.nbytes, tm.MEM_CPU)
    req._keep += [a, b, c]
    req.add_input("NUMKEYS", np.array([[batch * 2, batch * 26]], np.int32)).request_output()
    inst.execute([req])
    assert req.error_code == -1
    assert np.array_equal(_bits(req.output_numpy()), _bits(O.np_lookup(tables, q, [batch * 2, batch * 26], [0.0, 0.0])))


def test_empty_table_slices_are_legal(wdl_server):
    from oracle import hps_oracle as O
    srv, tables, _, _ = wdl_server
    inst = srv.load_model("hps_wdl", tm.model_config("hps_wdl", kind="KIND_CPU", gpus=[])).create_instance("i", tm.KIND_CPU)
    q = tables[1][0][:52].astype(np.int64)
    req = tm.Request("e").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.array([[0, 52]], np.int32)).request_output()
    req0 = tm.Request("z").add_input("KEYS", np.zeros((1, 0), np.int64)).add_input("NUMKEYS", np.array([[0, 0]], np.int32)).request_output()
    inst.execute([req, req0])
    assert req.error_code == -1 and req0.error_code == -1
    assert np.array_equal(_bits(req.output_numpy()), _bits(O.np_lookup(tables, q, [0, 52], [0.0, 0.0])))
    assert req0.output(0)[2] == [0]


def test_request_errors_become_error_responses_not_call_failures(wdl_server):
    srv, tables, _, _ = wdl_server
    inst = srv.load_model("hps_wdl", tm.model_config("hps_wdl", kind="KIND_CPU", gpus=[])).create_instance("i", tm.KIND_CPU)
    rng = np.random.default_rng(3)
    good, q, nk = _wdl_request(rng, tables, batch=4, rid="good")
    # (1) wrong input name (hps.cc:446-465 returns from Execute there; here: an error response)
    bad_name = tm.Request("bad_name").add_input("KEYZ", q.reshape(1, -1)).add_input("NUMKEYS", np.array([[8, 104]], np.int32)).request_output()
    # (2) more samples than max_batch_size (hps.cc:576-582): 1025 samples x 28 keys
    big = np.zeros((1, 1025 * 28), np.int64)
    too_big = tm.Request("too_big").add_input("KEYS", big).add_input("NUMKEYS", np.array([[1025 * 2, 1025 * 26]], np.int32)).request_output()
    # (3) NUMKEYS does not add up to the KEYS element count
    mismatch = tm.Request("mismatch").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.array([[8, 100]], np.int32)).request_output()
    # (4) NUMKEYS with the wrong number of tables
    wrong_t = tm.Request("wrong_t").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.array([[112]], np.int32)).request_output()
    # (5) wrong KEYS datatype
    wrong_dt = tm.Request("wrong_dt").add_input("KEYS", q.astype(np.int32).reshape(1, -1)).add_input("NUMKEYS", np.array([[8, 104]], np.int32)).request_output()
    # (6) negative count
    neg = tm.Request("neg").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.array([[-1, 113]], np.int32)).request_output()
    reqs = [bad_name, good, too_big, mismatch, wrong_t, wrong_dt, neg]
    inst.execute(reqs)  # the call itself succeeds
    for r in reqs:
        assert (r.response_count, r.release_count, r.final) == (1, 1, True), r
    assert good.error_code == -1
    assert bad_name.error_code == tm.ERR["INVALID_ARG"] and "KEYS and NUMKEYS" in bad_name.error_message
    assert too_big.error_code == tm.ERR["UNSUPPORTED"] and "more than max batch size" in too_big.error_message
    for r in (mismatch, wrong_t, wrong_dt, neg):
        assert r.error_code == tm.ERR["INVALID_ARG"], r.error_message
    st = inst.stats()
    assert (st.success_requests, st.failed_requests) == (1, 6)


def test_request_without_requested_output_gets_an_empty_success_response(wdl_server):
    srv, tables, _, _ = wdl_server
    inst = srv.load_model("hps_wdl", tm.model_config("hps_wdl", kind="KIND_CPU", gpus=[])).create_instance("i", tm.KIND_CPU)
    q = np.concatenate([tables[0][0][:2], tables[1][0][:26]]).astype(np.int64)
    req = tm.Request("noout").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.array([[2, 26]], np.int32))
    inst.execute([req])
    assert (req.response_count, req.error_code, req.output_count) == (1, -1, 0)  # hps.cc:555


@pytest.mark.parametrize("mutate,needle", [
    (lambda c: c["input"].pop(), "exactly the inputs KEYS and NUMKEYS"),
    (lambda c: c["input"][0].update(name="IDS"), "one KEYS and one NUMKEYS"),
    (lambda c: c["input"][0].update(data_type="TYPE_INT32"), "TYPE_INT64"),
    (lambda c: c["input"][1].update(data_type="TYPE_FP32"), "TYPE_INT32"),
    (lambda c: c["input"][0].update(dims=[26]), "variable first dimension"),
    (lambda c: c["output"].append(dict(c["output"][0])), "exactly one output"),
    (lambda c: c["output"][0].update(data_type="TYPE_FP16"), "TYPE_FP32"),
    (lambda c: c["output"][0].update(dims=[4]), "variable first dimension"),
    (lambda c: c.update(instance_group=[]), "at least one instance_group"),
    (lambda c: c["instance_group"][0].update(count=99), "num_of_worker_buffer_in_pool"),
])
def test_model_config_validation_rejects_what_the_reference_rejects(wdl_server, mutate, needle):
    """ModelState::CheckTensorContract / ReadDeployment: the accept/reject decisions of the reference's
    ValidateModelConfig / ParseModelConfig (model_state.cpp:180-371)."""
    srv, _, _, _ = wdl_server
    cfg = tm.model_config("hps_wdl", kind="KIND_CPU", gpus=[])
    mutate(cfg)
    with pytest.raises(tm.TritonError) as e:
        srv.load_model("hps_wdl", cfg)
    assert e.value.code == tm.ERR["INVALID_ARG"] and needle in e.value.msg


def test_unknown_model_is_not_found_instead_of_throwing(wdl_server):
    srv, _, _, _ = wdl_server
    with pytest.raises(tm.TritonError) as e:   # the reference's map.at() throws across the ABI (hps.cc:221-223)
        srv.load_model("not_in_ps_json", tm.model_config("not_in_ps_json", kind="KIND_CPU", gpus=[]))
    assert e.value.code == tm.ERR["NOT_FOUND"]


def test_online_deployment_of_a_model_added_to_ps_json_later(wdl_server):
    """A model absent at start-up is picked up by re-parsing ps.json in ModelInitialize (hps.cc:207-219)."""
    from oracle import hps_oracle as O
    srv, tables, ps_path, tmp = wdl_server
    new_tables = make_tables([(400, 8)], seed=5)
    d = tmp / "late0"
    O.np_write_table(d, *new_tables[0])
    cfg = json.loads(ps_path.read_text())
    cfg["models"].append(ps_config("late", new_tables, dirs=[str(d)], gpucache=False, maxcat=[4], defaults=[0.5])["models"][0])
    ps_path.write_text(json.dumps(cfg))
    model = srv.load_model("late", tm.model_config("late", kind="KIND_CPU", gpus=[]))
    inst = model.create_instance("late_0", tm.KIND_CPU)
    q = np.concatenate([new_tables[0][0][:30], [-9, -10]]).astype(np.int64)
    req = tm.Request("l").add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.array([[32]], np.int32)).request_output()
    inst.execute([req])
    assert req.error_code == -1
    assert np.array_equal(_bits(req.output_numpy()), _bits(O.np_lookup(new_tables, q, [32], [0.5])))


def test_backend_initialisation_failures(tmp_path):
    # no ps=... in the backend config
    with pytest.raises(tm.TritonError) as e:
        tm.Server(None, backend_config={"cmdline": {}})
    assert e.value.code == tm.ERR["INVALID_ARG"]
    # unreadable ps.json
    with pytest.raises(tm.TritonError):
        tm.Server(tmp_path / "missing.json")
    # Triton core older than the API the backend was built against (hps.cc:77-82)
    (tmp_path / "ps.json").write_text(json.dumps({"supportlonglong": True, "models": []}))
    with pytest.raises(tm.TritonError) as e:
        tm.Server(tmp_path / "ps.json", api=(1, 1))
    assert e.value.code == tm.ERR["UNSUPPORTED"]
    with pytest.raises(tm.TritonError) as e:
        tm.Server(tmp_path / "ps.json", api=(2, 99))
    assert e.value.code == tm.ERR["UNSUPPORTED"]
    srv = tm.Server(tmp_path / "ps.json")  # empty model list is only a warning (backend.cpp:311-316)
    srv.shutdown()


def test_two_models_two_instances_concurrently(tmp_path):
    """Different instances/models execute concurrently (hps.cc:353-359) on one shared parameter server."""
    import threading
    from oracle import hps_oracle as O
    all_tables, models = {}, []
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8}, "models": []}
    for m in ("wdl_a", "wdl_b"):
        tables = make_tables([(1500, 1), (1500, 16)], seed=hash(m) % 1000)
        dirs = []
        for t, (k, r) in enumerate(tables):
            d = tmp_path / f"{m}_{t}"
            O.np_write_table(d, k, r)
            dirs.append(str(d))
        cfg["models"].append(ps_config(m, tables, dirs=dirs, gpucache=False, maxcat=[2, 26], max_batch=1024)["models"][0])
        all_tables[m] = tables
    (tmp_path / "ps.json").write_text(json.dumps(cfg))
    srv = tm.Server(tmp_path / "ps.json")
    try:
        insts = {}
        for m in all_tables:
            mod = srv.load_model(m, tm.model_config(m, kind="KIND_CPU", gpus=[], count=2))
            insts[m] = [mod.create_instance(f"{m}_{i}", tm.KIND_CPU) for i in range(2)]
        errs = []

        def work(m, inst, seed):
            rng = np.random.default_rng(seed)
            for it in range(20):
                req, q, nk = _wdl_request(rng, all_tables[m], batch=int(rng.integers(1, 200)), rid=f"{m}{seed}{it}")
                inst.execute([req])
                ref = O.np_lookup(all_tables[m], q, nk, [0.0, 0.0])
                if req.error_code != -1 or not np.array_equal(_bits(req.output_numpy()), _bits(ref)):
                    errs.append((m, seed, it, req.error_message))
        th = [threading.Thread(target=work, args=(m, inst, 10 * j + i)) for j, m in enumerate(all_tables) for i, inst in enumerate(insts[m])]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs[:3]
    finally:
        srv.shutdown()


def test_host_tier_smaller_than_the_table_through_the_plugin(tmp_path):
    """ps.json with the reference's overflow keys and a persistent database (docs/hierarchical_parameter_server.md:460-569):
    the plugin serves exact rows while the bounded volatile tier prunes underneath (tests/test_host_tier_bounded.py has
    the tier's own checks); the store directory is created under persistent_db.path."""
    from oracle import hps_oracle as O
    tables = make_tables([(3000, 1), (2000, 16)])
    dirs = []
    for t, (k, r) in enumerate(tables):
        O.np_write_table(tmp_path / f"t{t}", k, r)
        dirs.append(str(tmp_path / f"t{t}"))
    cfg = ps_config("hps_wdl", tables, dirs=dirs, gpucache=False, maxcat=[2, 26], defaults=[0.0, 0.0], max_batch=1024)
    cfg["volatile_db"].update({"overflow_margin": 40, "overflow_policy": "evict_least_used", "overflow_resolution_target": 0.5,
                               "initial_cache_rate": 0.1, "cache_missed_embeddings": True})
    cfg["persistent_db"] = {"type": "rocks_db", "path": str(tmp_path / "pdb"), "num_threads": 4, "read_only": False}
    (tmp_path / "ps.json").write_text(json.dumps(cfg))
    srv = tm.Server(tmp_path / "ps.json")
    try:
        inst = srv.load_model("hps_wdl", tm.model_config("hps_wdl", kind="KIND_CPU", gpus=[])).create_instance("i0", tm.KIND_CPU, 0)
        rng = np.random.default_rng(3)
        for i in range(30):
            req, q, nk = _wdl_request(rng, tables, batch=int(rng.integers(1, 30)), rid=str(i))
            inst.execute([req])
            assert req.error_code == -1
            assert np.array_equal(_bits(req.output_numpy()), _bits(O.np_lookup(tables, q, nk, [0.0, 0.0])))
            req.close()
        assert (tmp_path / "pdb" / "hps_wdl" / "sparse_embedding2" / "emb_vector").stat().st_size == tables[1][1].nbytes
    finally:
        srv.shutdown()


def test_native_driver_config1_cpu_parameter_server_through_the_plugin():
    """BASELINE configs[0] — the reference's CI smoke is perf_analyzer against a CPU-only deployment (.gitlab-ci.yml:70): the
    native load generator (tools/triton_abi_bench.cpp) plays that role here: 1 table x 16 floats, 4,096-key requests,
    gpucache=false, KIND_CPU instance, OUTPUT0 in host memory, every sampled row checked against the table recipe; and a
    two-table W&D-shaped CPU model (D = [1,16], keys per sample [2,26]) for the per-table slicing of NUMKEYS / OUTPUT0."""
    import json
    import subprocess
    from hugectr_backend_amd import build as hb
    exe = hb.LIB / "triton_abi_bench.bin"
    assert exe.exists()
    for extra, nkeys, nout in (
            (["--tables", "1", "--rows", "200000", "--dims", "16", "--batch", "4096", "--uniform", "1"], 4096, 4096 * 16),
            (["--models", "2", "--rows", "50000", "--dims", "1,16", "--per-sample", "2,26", "--batch", "64"], 64 * 28, 64 * (2 + 26 * 16))):
        r = subprocess.run([str(exe), "--lib-dir", str(hb.LIB), *extra, "--gpucache", "0", "--instances", "1", "--steps", "20",
                            "--blocks", "2", "--warmup", "5"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert d["failed"] == 0 and d["rows_wrong"] == 0 and d["rows_checked_against_recipe"] >= 2048
        assert d["keys_per_request"] == nkeys and d["floats_per_response"] == nout and d["output_memory"] == "host"
        assert d["requests_ok_reported_by_backend"] == d["batch_statistics_reports"] == 45
    # --requests-per-execute 4: four requests per TRITONBACKEND_ModelInstanceExecute call (a dynamic batcher's hand-over), served
    # by one lookup each time; the LAST request's rows of the last call are the ones checked against the recipe
    r = subprocess.run([str(exe), "--lib-dir", str(hb.LIB), "--models", "2", "--rows", "50000", "--dims", "1,16", "--per-sample", "2,26",
                        "--batch", "64", "--gpucache", "0", "--instances", "1", "--steps", "10", "--blocks", "2", "--warmup", "3",
                        "--requests-per-execute", "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["failed"] == 0 and d["rows_wrong"] == 0 and d["requests_per_execute"] == 4
    assert d["requests_ok_reported_by_backend"] == 4 * 23 and d["batch_statistics_reports"] == 23
    assert d["instances_whose_last_execute_was_one_lookup"] == 2
