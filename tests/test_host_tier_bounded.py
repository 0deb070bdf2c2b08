"""Host tier smaller than the table: bounded volatile database (overflow_margin / overflow_policy /
overflow_resolution_target / initial_cache_rate / cache_missed_embeddings) in front of a persistent row store
(persistent_db) — docs/hierarchical_parameter_server.md:460-569 of the reference, SURVEY.md §8(f) rank 3.
Checked against oracle.hps_oracle.VolatileDbModel (content of the tier) and np_lookup (the rows served)."""
import os

import numpy as np
import pytest

from tests.conftest import make_tables, ps_config


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _server(tmp_path, tables, vdb=None, pdb=None, partitions=8, write_files=True, **kw):
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    dirs = []
    for t, (k, r) in enumerate(tables):
        d = tmp_path / f"model_t{t}"
        if write_files:
            O.np_write_table(d, k, r)
        dirs.append(str(d))
    cfg = ps_config("m", tables, dirs=dirs, gpucache=kw.pop("gpucache", False), **kw)
    cfg["volatile_db"].update({"num_partitions": partitions, **(vdb or {})})
    if pdb is not None:
        cfg["persistent_db"] = {"type": "rocks_db", "path": str(tmp_path / "store"), **pdb}
    return hps.HierParameterServer.create_from_dict(cfg, load_tables=True)


def _queries(rng, keys, n, cold=0.1):
    q = rng.choice(keys, n)
    m = rng.random(n) < cold
    q[m] = rng.integers(10**12, 10**13, int(m.sum()))
    return q.astype(np.int64)


@pytest.mark.parametrize("policy", ["evict_oldest", "evict_least_used"])
@pytest.mark.parametrize("partitions", [1, 8, 5])
def test_bounded_tier_content_follows_the_model(tmp_path, policy, partitions):
    """Every fetch returns the exact rows (whatever tier serves them) and the tier's content after every call is the
    model's: same keys cached initially, same victims at every overflow."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(partitions * 7 + len(policy))
    tables = make_tables([(3000, 8)])
    k, r = tables[0]
    vdb = {"overflow_margin": 100, "overflow_policy": policy, "overflow_resolution_target": 0.7,
           "initial_cache_rate": 0.1, "cache_missed_embeddings": True}
    ps = _server(tmp_path, tables, vdb=vdb, pdb={}, partitions=partitions, defaults=[0.25])
    model = O.VolatileDbModel(k, partitions, 100, policy, 0.7, 0.1, True)
    assert np.array_equal(ps.host_tier_keys("m", 0), model.resident())
    hot = rng.choice(k, 400, replace=False)
    for it in range(60):
        q = _queries(rng, hot if it % 3 else k, 48)     # <= 64 keys: one fetch task, one access stamp per call
        out, found = ps.fetch("m", 0, q, return_found=True)
        ref = O.np_lookup([(k, r)], q, [q.size], [0.25]).reshape(q.size, -1)
        assert np.array_equal(_bits(out), _bits(ref))
        assert np.array_equal(found.astype(bool), model.fetch(q))
        assert np.array_equal(ps.host_tier_keys("m", 0), model.resident()), f"call {it}"
    st = ps.host_tier_stats("m", 0)
    assert st["tiered"] == 1 and st["persistent_rows"] == k.size
    assert st["max_partition_entries"] <= 100 and st["entries"] == model.resident().size
    assert st["evictions"] == model.evictions and st["overflows"] > 0
    assert st["hits"] + st["persistent_hits"] + st["not_found"] == 60 * 48


def test_evict_random_respects_margin_and_target(tmp_path):
    from oracle import hps_oracle as O
    rng = np.random.default_rng(3)
    tables = make_tables([(4000, 4)])
    k, r = tables[0]
    vdb = {"overflow_margin": 64, "overflow_policy": "evict_random", "overflow_resolution_target": 0.5,
           "initial_cache_rate": 0.0, "cache_missed_embeddings": True}
    ps = _server(tmp_path, tables, vdb=vdb, pdb={}, partitions=4)
    assert ps.host_tier_keys("m", 0).size == 0
    prev = ps.host_tier_stats("m", 0)
    for _ in range(80):
        q = rng.choice(k, 40).astype(np.int64)
        out = ps.fetch("m", 0, q)
        assert np.array_equal(_bits(out), _bits(O.np_lookup([(k, r)], q, [q.size], [0.0]).reshape(q.size, -1)))
        st = ps.host_tier_stats("m", 0)
        assert st["max_partition_entries"] <= 64
        # each overflow prunes one partition from 64 to 32 entries
        assert st["evictions"] - prev["evictions"] == 32 * (st["overflows"] - prev["overflows"])
        held = ps.host_tier_keys("m", 0)
        assert np.isin(held, k).all() and held.size == st["entries"] == st["inserts"] - st["evictions"]
        prev = st
    assert prev["overflows"] >= 4


def test_without_persistent_db_the_bounded_tier_is_the_database(tmp_path):
    """No persistent database: what the volatile tier does not hold is not found (default vector), as in the
    reference, where pruned embeddings are gone."""
    from oracle import hps_oracle as O
    tables = make_tables([(1000, 6)])
    k, r = tables[0]
    vdb = {"overflow_margin": 50, "overflow_policy": "evict_oldest", "initial_cache_rate": 1.0}
    ps = _server(tmp_path, tables, vdb=vdb, partitions=4, defaults=[-1.0])
    model = O.VolatileDbModel(k, 4, 50, "evict_oldest", 0.8, 1.0, False, persistent=False)
    held = ps.host_tier_keys("m", 0)
    assert np.array_equal(held, model.resident()) and 0 < held.size <= 200
    out, found = ps.fetch("m", 0, k, return_found=True)
    assert np.array_equal(found.astype(bool), np.isin(k, held))
    ref = O.np_lookup([(k, r)], k, [k.size], [-1.0], resident=[held]).reshape(k.size, -1)
    assert np.array_equal(_bits(out), _bits(ref))
    # an online update lands in the tier (and may prune it)
    nk = np.array([k[0], 10**15 + 3], dtype=np.int64)
    nr = np.full((2, 6), 7.5, np.float32)
    ps.upsert("m", 0, nk, nr)
    out, found = ps.fetch("m", 0, nk, return_found=True)
    assert found.all() and np.array_equal(out, nr)
    assert ps.host_tier_stats("m", 0)["max_partition_entries"] <= 50


def test_persistent_store_survives_and_read_only_reuses_it(tmp_path):
    from oracle import hps_oracle as O
    tables = make_tables([(500, 4), (300, 2)])
    vdb = {"overflow_margin": 16, "initial_cache_rate": 0.2, "cache_missed_embeddings": False}
    ps = _server(tmp_path, tables, vdb=vdb, pdb={"read_only": False})
    for t in range(2):
        # <path>/<model>/<table name>/, default table names sparse_embedding1..T
        assert (tmp_path / "store" / "m" / f"sparse_embedding{t + 1}" / "emb_vector").stat().st_size == tables[t][1].nbytes
    # write-through update: an existing key and a new one
    k0, r0 = tables[0]
    nk = np.array([k0[5], 10**14], dtype=np.int64)
    nr = np.stack([np.full(4, 2.0, np.float32), np.full(4, 3.0, np.float32)])
    ps.upsert("m", 0, nk, nr)
    assert np.array_equal(ps.fetch("m", 0, nk), nr)
    before = ps.host_tier_keys("m", 0)
    assert np.array_equal(ps.fetch("m", 0, k0[100:140]), r0[100:140])
    assert np.array_equal(ps.host_tier_keys("m", 0), before)        # cache_missed_embeddings = false: content unchanged
    ps.close()
    # the model files disappear; a read_only deployment serves everything from the store, including the update
    for t in range(2):
        for f in ("key", "emb_vector"):
            os.remove(tmp_path / f"model_t{t}" / f)
    ps2 = _server(tmp_path, tables, vdb=vdb, pdb={"read_only": True}, write_files=False)
    exp = r0.copy()
    exp[5] = 2.0
    assert np.array_equal(ps2.fetch("m", 0, k0), exp)
    assert np.array_equal(ps2.fetch("m", 0, nk[1:]), nr[1:])
    assert np.array_equal(ps2.fetch("m", 1, tables[1][0]), tables[1][1])
    with pytest.raises(Exception, match="read_only"):
        ps2.upsert("m", 0, np.array([10**14 + 1], np.int64), np.zeros((1, 4), np.float32))
    # an update of an existing key on a read_only store refreshes a cached copy only
    ps2.upsert("m", 0, k0[:1], np.full((1, 4), 9.0, np.float32))
    ps2.close()


def test_duplicate_keys_in_the_files_and_in_a_request(tmp_path):
    from oracle import hps_oracle as O
    rng = np.random.default_rng(5)
    base = rng.choice(10**6, 300, replace=False).astype(np.int64)
    k = np.concatenate([base, base[:50]])                      # 50 keys appear twice: the later row is the live one
    r = rng.random((k.size, 3), dtype=np.float32)
    vdb = {"overflow_margin": 20, "overflow_policy": "evict_least_used", "initial_cache_rate": 1.0,
           "cache_missed_embeddings": True}
    ps = _server(tmp_path, [(k, r)], vdb=vdb, pdb={}, partitions=2)
    model = O.VolatileDbModel(k, 2, 20, "evict_least_used", 0.8, 1.0, True)
    assert np.array_equal(ps.host_tier_keys("m", 0), model.resident())
    q = np.concatenate([base[:30], base[:30], base[40:60]])    # duplicates inside the request
    out = ps.fetch("m", 0, q)
    assert np.array_equal(_bits(out), _bits(O.np_lookup([(k, r)], q, [q.size], [0.0]).reshape(q.size, -1)))
    model.fetch(q)
    assert np.array_equal(ps.host_tier_keys("m", 0), model.resident())


def test_unbounded_persistent_config_keeps_the_fast_path(tmp_path):
    """persistent_db enabled with the default (unlimited) volatile database: the table is served from RAM as before,
    the store is still materialised."""
    tables = make_tables([(200, 4)])
    ps = _server(tmp_path, tables, pdb={})
    st = ps.host_tier_stats("m", 0)
    assert st["tiered"] == 0 and st["persistent_rows"] == 200
    assert np.array_equal(ps.fetch("m", 0, tables[0][0]), tables[0][1])
    assert (tmp_path / "store" / "m").is_dir()


def test_configuration_errors(tmp_path):
    from hugectr_backend_amd import hps
    tables = make_tables([(50, 2)])
    cfg = ps_config("m", tables, gpucache=False)
    for bad, msg in (({"initial_cache_rate": 1.5}, "initial_cache_rate"), ({"overflow_margin": 0}, "overflow_margin")):
        c = {**cfg, "volatile_db": {**cfg["volatile_db"], **bad}}
        with pytest.raises(Exception, match=msg):
            hps.HierParameterServer.create_from_dict(c, load_tables=False)
    c = {**cfg, "persistent_db": {"type": "rocks_db", "path": str(tmp_path / "nowhere"), "read_only": True}}
    c["models"][0]["sparse_files"] = [str(tmp_path / "absent")]
    with pytest.raises(Exception, match="cannot stat"):
        hps.HierParameterServer.create_from_dict(c, load_tables=True)


def test_concurrent_fetches_with_pruning_return_exact_rows(tmp_path):
    """Four threads hammer a tiny tier (constant pruning) through large requests that fan out over the serving pool."""
    import threading
    from oracle import hps_oracle as O
    tables = make_tables([(20000, 16)])
    k, r = tables[0]
    vdb = {"overflow_margin": 200, "overflow_policy": "evict_oldest", "initial_cache_rate": 0.05,
           "cache_missed_embeddings": True}
    ps = _server(tmp_path, tables, vdb=vdb, pdb={})
    errs = []

    def worker(seed):
        rng = np.random.default_rng(seed)
        for _ in range(20):
            q = _queries(rng, k, 3000, cold=0.05)
            out = ps.fetch("m", 0, q)
            ref = O.np_lookup([(k, r)], q, [q.size], [0.0]).reshape(q.size, -1)
            if not np.array_equal(_bits(out), _bits(ref)):
                errs.append(seed)

    ts = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    st = ps.host_tier_stats("m", 0)
    assert st["max_partition_entries"] <= 200 and st["overflows"] > 0
    assert st["hits"] + st["persistent_hits"] + st["not_found"] == 4 * 20 * 3000


def test_documented_overflow_example(tmp_path):
    """docs/hierarchical_parameter_server.md:487-489: "The default value is 0.8 and indicates to evict embeddings from a
    partition until it is shrunk to 80% of its maximum size. In other words, when the partition size surpasses
    overflow_margin embeddings, 20% of the embeddings are evicted according to the specified overflow_policy."""
    tables = make_tables([(100, 2)])
    k, _ = tables[0]
    vdb = {"overflow_margin": 10, "overflow_policy": "evict_oldest", "initial_cache_rate": 0.0,
           "cache_missed_embeddings": True}                      # overflow_resolution_target left at its default
    ps = _server(tmp_path, tables, vdb=vdb, pdb={}, partitions=1)
    for i in range(10):
        ps.fetch("m", 0, k[i:i + 1])
    assert ps.host_tier_stats("m", 0)["entries"] == 10           # at the margin: nothing evicted yet
    ps.fetch("m", 0, k[10:11])                                   # surpasses it: 20 % go (the two oldest), the new one enters
    st = ps.host_tier_stats("m", 0)
    assert (st["entries"], st["evictions"], st["overflows"]) == (9, 2, 1)
    assert np.array_equal(ps.host_tier_keys("m", 0), np.sort(k[2:11]))


@pytest.mark.parametrize("persistent", [False, True])
def test_margin_above_the_load_time_key_count_takes_new_keys(tmp_path, persistent):
    """Round-1 advisor finding: with overflow_margin >= the keys a partition held at load time (the reference's own
    example is overflow_margin = 10,000,000) a fully cached partition was "full", and the first NEW key — an online
    update into a volatile-only tier, or a row appended to the store and then cached on its first fetch — popped an
    empty free list (heap corruption).  The margin is the limit; storage grows on demand."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(5)
    tables = make_tables([(1000, 8)])
    k, r = tables[0]
    vdb = {"overflow_margin": 10_000_000, "overflow_policy": "evict_oldest", "initial_cache_rate": 1.0,
           "cache_missed_embeddings": True}
    ps = _server(tmp_path, tables, vdb=vdb, pdb={} if persistent else None, partitions=4, defaults=[0.5])
    assert ps.host_tier_stats("m", 0)["entries"] == 1000
    new_k = np.arange(10**9, 10**9 + 300, dtype=np.int64)
    new_r = rng.random((300, 8), dtype=np.float32)
    for lo in range(0, 300, 64):                       # several rounds: the slabs have to grow more than once
        ps.upsert("m", 0, new_k[lo:lo + 64], new_r[lo:lo + 64])
    # volatile-only: the tier is the database and took the rows; persistent: they went to the store
    assert ps.host_tier_stats("m", 0)["entries"] == (1000 if persistent else 1300)
    q = np.concatenate([new_k, k[::7], [42 + 10**12]]).astype(np.int64)
    for _ in range(2):                                 # the first fetch caches what came from the store
        out, found = ps.fetch("m", 0, q, return_found=True)
        ref = O.np_lookup([(np.concatenate([k, new_k]), np.concatenate([r, new_r]))], q, [q.size], [0.5]).reshape(q.size, -1)
        assert np.array_equal(_bits(out), _bits(ref))
        assert found[:-1].all() and not found[-1]
    st = ps.host_tier_stats("m", 0)
    assert st["entries"] == 1300 and st["evictions"] == 0
    assert np.array_equal(ps.host_tier_keys("m", 0), np.sort(np.concatenate([k, new_k])))


def test_margin_of_one_evicts_one_entry_per_insert(tmp_path):
    """keep = max(1, floor(1 * 0.8)) = 1 = the margin: the prune frees nothing by itself, the insert still needs room."""
    tables = make_tables([(50, 2)])
    k, _ = tables[0]
    vdb = {"overflow_margin": 1, "overflow_policy": "evict_oldest", "initial_cache_rate": 0.0, "cache_missed_embeddings": True}
    ps = _server(tmp_path, tables, vdb=vdb, pdb={}, partitions=1)
    for i in range(5):
        ps.fetch("m", 0, k[i:i + 1])
        assert np.array_equal(ps.host_tier_keys("m", 0), k[i:i + 1])
    assert ps.host_tier_stats("m", 0)["evictions"] == 4
