"""Host (CPU RAM) tier of the parameter server — the `hash_map` volatile database of the reference
(docs/hierarchical_parameter_server.md:380-412) — against the oracle.  This is BASELINE config 1's path."""
import numpy as np
import pytest

from tests.conftest import make_tables, ps_config


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _server(tables, partitions=8, **kw):
    from hugectr_backend_amd import hps
    cfg = ps_config("m", tables, gpucache=False, **kw)
    cfg["volatile_db"]["num_partitions"] = partitions
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays("m", t, k, r)
    return ps


@pytest.mark.parametrize("partitions", [1, 3, 8, 64])
def test_fetch_matches_oracle_for_any_partition_count(partitions):
    from oracle import hps_oracle as O
    rng = np.random.default_rng(partitions)
    tables = make_tables([(20000, 16), (777, 5)])
    ps = _server(tables, partitions, defaults=[0.5, -2.0])
    for t, (k, r) in enumerate(tables):
        q = np.concatenate([rng.choice(k, 5000), rng.integers(10**9, 10**10, 500)]).astype(np.int64)
        rng.shuffle(q)
        out, found = ps.fetch("m", t, q, return_found=True)
        ref = O.np_lookup([(k, r)], q, [q.size], [[0.5, -2.0][t]]).reshape(q.size, -1)
        assert np.array_equal(_bits(out), _bits(ref))
        assert np.array_equal(found.astype(bool), np.isin(q, k))


def test_config1_shape_single_table_4k_batch():
    """BASELINE config 1 at reduced rows: one table x 16-dim, 4,096-key batch, CPU parameter server only."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    R = 1 << 16
    keys = np.arange(R, dtype=np.int64)
    rows = O.c_synth_rows(O.SEED, 0, 0, R, 16)
    ps = _server([(keys, rows)], max_batch=4096)
    s = hps.LookupSession.create(ps, "m", None)
    rng = np.random.default_rng(0)
    co = O.COracle()
    co.add_table_arrays(keys, rows)
    for _ in range(5):
        q = rng.integers(0, R, 4096).astype(np.int64)
        assert np.array_equal(_bits(s.lookup(q, [4096])), _bits(co.lookup(q, [4096], [0.0])))


def test_synthetic_table_loader_matches_oracle_generator():
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    tables = [(np.arange(10), np.zeros((10, 128), np.float32)), (np.arange(10), np.zeros((10, 3), np.float32))]
    ps = hps.HierParameterServer.create_from_dict(ps_config("m", tables, gpucache=False), load_tables=False)
    ps.load_table_synthetic("m", 0, O.SEED, 100, 5000)
    ps.load_table_synthetic("m", 1, O.SEED, 0, 333)
    q = np.array([100, 101, 5099, 5100, 99, 2500], np.int64)
    out, found = ps.fetch("m", 0, q, return_found=True)
    assert found.tolist() == [1, 1, 1, 0, 0, 1]
    ref = O.np_synth_rows(O.SEED, 0, q, 128)
    assert np.array_equal(_bits(out[found.astype(bool)]), _bits(ref[found.astype(bool)]))
    out1 = ps.fetch("m", 1, np.arange(333, dtype=np.int64))
    assert np.array_equal(_bits(out1), _bits(O.np_synth_rows(O.SEED, 1, np.arange(333), 3)))  # odd D
    assert ps.table_info("m", 0).rows_loaded == 5000


def test_synthetic_shard_loader_partitions_the_table_by_owner():
    """One rank's slice of a model-parallel table (BASELINE config 3): exactly the keys mix64(key) mod P assigns to it,
    with the rows of the unsharded recipe."""
    from hugectr_backend_amd import hps
    from hugectr_backend_amd.sharded import owner_of
    from oracle import hps_oracle as O
    P, R, key0, D = 3, 70001, 17, 20
    tables = [(np.arange(10), np.zeros((10, D), np.float32))]
    allk = np.arange(key0, key0 + R, dtype=np.int64)
    own = owner_of(allk, P)
    total = 0
    for rank in range(P):
        ps = hps.HierParameterServer.create_from_dict(ps_config(f"s{rank}", tables, gpucache=False), load_tables=False)
        ps.load_table_synthetic(f"s{rank}", 0, O.SEED, key0, R, shard=rank, num_shards=P)
        n = ps.table_info(f"s{rank}", 0).rows_loaded
        assert n == int((own == rank).sum())
        total += n
        out, found = ps.fetch(f"s{rank}", 0, allk, return_found=True)
        assert np.array_equal(found.astype(bool), own == rank)
        mine = allk[own == rank]
        assert np.array_equal(_bits(out[own == rank]), _bits(O.np_synth_rows(O.SEED, 0, mine, D)))
    assert total == R
    with pytest.raises(hps.HpsError):
        ps.load_table_synthetic(f"s{P - 1}", 0, O.SEED, 0, 10, shard=5, num_shards=3)


def test_duplicate_keys_last_wins_and_sentinel_key():
    from oracle import hps_oracle as O
    k = np.array([9, 4, 9, 7, 4, np.iinfo(np.int64).min, 9], np.int64)
    r = O.np_synth_rows(1, 0, np.arange(k.size), 4)
    ps = _server([(k, r)], defaults=[3.0])
    q = np.array([9, 4, 7, np.iinfo(np.int64).min, 1], np.int64)
    out = ps.fetch("m", 0, q)
    assert np.array_equal(_bits(out), _bits(np.stack([r[6], r[4], r[3], r[5], np.full(4, 3.0, np.float32)])))


def test_host_tier_lookup_errors():
    from hugectr_backend_amd import hps
    tables = make_tables([(100, 4), (100, 4)])
    ps = _server(tables, maxcat=[1, 1], max_batch=16)
    s = hps.LookupSession.create(ps, "m", None)
    with pytest.raises(hps.HpsError) as e:
        s.lookup_ptrs([0], [0], [1])  # wrong table count
    assert e.value.code == hps.ERR_INVALID_ARG
    with pytest.raises(hps.HpsError):
        s.lookup(np.arange(5), [2, 2])  # sum(NUMKEYS) != len(KEYS)
    with pytest.raises(hps.HpsError) as e:
        hps.LookupSession.create(ps, "ghost", None)
    assert e.value.code == hps.ERR_NOT_FOUND


def test_cache_handle_of_a_cpu_only_model_creates_its_session():
    """The shell asks get_embedding_cache(model, device) for every model and hands the result to
    LookupSessionBase::create(params, cache) (model_instance_state.cpp:168-171): a model without GPU cache gets a handle
    that knows its server and model, reports the table count and opens a host-tier session."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    tables = make_tables([(500, 4), (300, 8)])
    ps = hps.HierParameterServer.create_from_dict(ps_config("cpuonly", tables, gpucache=False, maxcat=[1, 2], defaults=[0.5, 1.5]),
                                                  load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays("cpuonly", t, k, r)
    cache = ps.get_embedding_cache("cpuonly", 0)
    assert cache is not None and cache.on_device is False and cache.num_tables == 2
    with pytest.raises(hps.HpsError):
        cache.counters()
    assert ps.get_embedding_cache("no_such_model", 0) is None
    s = hps.LookupSession.create_from_cache(ps, "cpuonly", cache)
    q = np.concatenate([tables[0][0][:10], [-5], tables[1][0][:20]]).astype(np.int64)
    out = s.lookup(q, [11, 20])
    assert np.array_equal(out.view(np.uint32), O.np_lookup(tables, q, [11, 20], [0.5, 1.5]).view(np.uint32))


def test_synthetic_table_source_and_table_data_view(tmp_path):
    """sparse_files entries of the form synthetic://<rows>[?seed=..&key0=..] load the SURVEY 8(d) recipe table instead of a
    directory (how a benchmark puts a full-size model behind the plugin boundary); hps_server_table_data hands out the
    table as it sits in the host tier without a copy.  Rows must equal the ORACLE's restatement of the recipe."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    R, D = 5000, 16
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
           "models": [{"model": "syn", "sparse_files": [f"synthetic://{R}", f"synthetic://{R // 2}?seed=7&key0=1000"],
                       "num_of_worker_buffer_in_pool": 1, "embedding_vecsize_per_table": [D, 4],
                       "maxnum_catfeature_query_per_table_per_sample": [1, 1], "default_value_for_each_table": [0.0, 9.0],
                       "deployed_device_list": [0], "max_batch_size": 4096, "gpucache": False}]}
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=True)
    k0, r0 = ps.table_data("syn", 0)
    k1, r1 = ps.table_data("syn", 1)
    assert np.array_equal(k0, np.arange(R)) and np.array_equal(k1, 1000 + np.arange(R // 2))
    assert np.array_equal(r0.view(np.uint32), O.np_synth_rows(O.SEED, 0, k0, D).view(np.uint32))
    assert np.array_equal(r1.view(np.uint32), O.np_synth_rows(7, 1, k1, 4).view(np.uint32))
    s = hps.LookupSession.create(ps, "syn", None)
    q = np.array([0, R - 1, R, 1000, 999, 1000 + R // 2 - 1], dtype=np.int64)
    out = s.lookup(q, [3, 3])
    exp = np.concatenate([O.np_synth_rows(O.SEED, 0, q[:2], D).ravel(), np.zeros(D, np.float32),
                          O.np_synth_rows(7, 1, q[3:4], 4).ravel(), np.full(4, 9.0, np.float32),
                          O.np_synth_rows(7, 1, q[5:6], 4).ravel()])
    assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))
    # a malformed source is a directory name, and that directory does not exist
    cfg["models"][0]["sparse_files"][0] = "synthetic://12abc"
    with pytest.raises(hps.HpsError):
        hps.HierParameterServer.create_from_dict(cfg, load_tables=True)


def test_copy_engine_wakeup_is_skipped_without_a_device():
    from hugectr_backend_amd import hps
    if hps.device_count() > 0:
        pytest.skip("a GPU is present")
    n, report = hps.wake_copy_engines(0)
    assert n == 0 and report.startswith("skipped"), report
