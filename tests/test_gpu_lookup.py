"""GPU parity tests: HIP lookup path (through the C ABI of libhps_amd.so) vs the CPU oracle.

Bit-exact bar: the path only moves fp32 rows, so every comparison is on the uint32 view.
Run on the MI355X box with `pytest -m gpu`.
"""
import numpy as np
import pytest

from tests.conftest import make_tables, ps_config

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _mk(model, tables, **kw):
    """server + cache + session for one model whose tables are injected from arrays."""
    from hugectr_backend_amd import hps
    cfg = ps_config(model, tables, **kw)
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays(model, t, k, r)
    cache = None
    if cfg["models"][0]["gpucache"]:
        ps.create_embedding_cache_per_model(model)
        cache = ps.get_embedding_cache(model, 0)
        assert cache is not None
    sess = hps.LookupSession.create(ps, model, cache)
    return ps, cache, sess


def _queries(rng, tables, num_keys, miss_frac=0.2):
    """per table: mostly existing keys (with duplicates), some keys that exist nowhere."""
    parts = []
    for (keys, _), n in zip(tables, num_keys):
        q = rng.choice(keys, size=n, replace=True)
        absent = rng.random(n) < miss_frac
        q = np.where(absent, -1 - rng.integers(0, 1 << 40, n), q)  # negative keys are never in the tables
        parts.append(q.astype(np.int64))
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


@pytest.mark.usefixtures("plain_lru")   # watches the insert mechanics: every missed key is taken in (conftest.py)
def test_wdl_shape_sync_exact():
    """W&D request shape of the reference sample: D=[1,16], 10 samples, keys/sample [2,26] -> 4180 floats
    (samples/Hierarchical_Parameter_Server_Deployment.ipynb:738-747,793-795)."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(1)
    tables = make_tables([(3000, 1), (2000, 16)])
    ps, cache, s = _mk("wdl", tables, maxcat=[2, 26], defaults=[0.0, 0.0], gpucacheper=0.5)
    nk = [20, 260]
    q = _queries(rng, tables, nk)
    out = s.lookup(q, nk).cpu().numpy()
    assert out.shape == (4180,)
    ref = O.np_lookup(tables, q, nk, [0.0, 0.0])
    assert np.array_equal(_bits(out), _bits(ref))
    st = s.last_stats()
    assert st.async_insert == 0
    # second identical call: everything that exists is now cached (sync insert) -> only absent keys miss
    out2 = s.lookup(q, nk).cpu().numpy()
    assert np.array_equal(_bits(out2), _bits(ref))
    st2 = s.last_stats()
    n_absent = int((q < 0).sum())
    # (a key whose bucket was full of the current recency unit's keys stays uncached: a handful at most)
    assert n_absent <= st2.misses <= n_absent + 8


@pytest.mark.parametrize("D", [1, 3, 4, 16, 32, 100, 128, 256])
def test_dims_sync_exact(D):
    from oracle import hps_oracle as O
    rng = np.random.default_rng(D)
    tables = make_tables([(5000, D), (777, D)])
    ps, cache, s = _mk(f"m{D}", tables, maxcat=[3, 2], defaults=[0.25, -1.0], gpucacheper=0.3, max_batch=2048)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    for it in range(3):
        nk = [int(rng.integers(0, 6000)), int(rng.integers(0, 4000))]
        q = _queries(rng, tables, nk)
        out = s.lookup(q, nk).cpu().numpy()
        ref = co.lookup(q, nk, [0.25, -1.0])
        assert np.array_equal(_bits(out), _bits(ref)), f"iter {it}"


def test_empty_and_ragged_tables():
    from oracle import hps_oracle as O
    rng = np.random.default_rng(7)
    tables = make_tables([(100, 8), (200, 8), (300, 8), (50, 8)])
    ps, cache, s = _mk("ragged", tables, maxcat=[4, 4, 4, 4], gpucacheper=1.0, max_batch=512)
    for nk in ([0, 0, 0, 0], [0, 5, 0, 1], [64, 0, 1, 0], [1, 1, 1, 1], [63, 65, 129, 1], [2048, 0, 0, 2048]):
        q = _queries(rng, tables, nk)
        out = s.lookup(q, nk).cpu().numpy()
        ref = O.np_lookup(tables, q, nk, [0.0] * 4)
        assert out.shape == ref.shape
        assert np.array_equal(_bits(out), _bits(ref)), nk


def test_criteo_shape_26x128_sync_exact_and_hits():
    """26 tables x D=128 (BASELINE config 2 shape at reduced rows): exact rows, warm cache hit accounting."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(26)
    T, R, D, B = 26, 20000, 128, 4096
    tables = make_tables([(R, D)] * T)
    ps, cache, s = _mk("criteo", tables, maxcat=[1] * T, gpucacheper=0.2, max_batch=B)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    nk = [B] * T
    # the cache was warmed with the first 20 % of each table in file order
    parts = []
    for keys, _ in tables:
        hot = keys[: int(0.2 * R)]
        cold = keys[int(0.2 * R):]
        pick_hot = rng.random(B) < 0.95
        parts.append(np.where(pick_hot, rng.choice(hot, B), rng.choice(cold, B)))
    q = np.concatenate(parts).astype(np.int64)
    out = s.lookup(q, nk).cpu().numpy()
    ref = co.lookup(q, nk, [0.0] * T, threads=4)
    assert np.array_equal(_bits(out), _bits(ref))
    st = s.last_stats()
    hit = 1.0 - st.misses / q.size
    assert 0.90 < hit < 0.97, hit  # warm-up may drop a few keys of over-full buckets
    assert st.unique_misses <= st.misses


def test_device_resident_keys_path():
    import torch
    from oracle import hps_oracle as O
    rng = np.random.default_rng(3)
    tables = make_tables([(4000, 128), (4000, 128), (1000, 16)])
    ps, cache, s = _mk("dev", tables, maxcat=[2, 2, 1], gpucacheper=0.5, max_batch=4096)
    nk = [5000, 3000, 777]
    q = _queries(rng, tables, nk, miss_frac=0.05)
    dq = torch.from_numpy(q).cuda()
    out = s.lookup_device(dq, nk).cpu().numpy()
    ref = O.np_lookup(tables, q, nk, [0.0] * 3)
    assert np.array_equal(_bits(out), _bits(ref))


@pytest.mark.usefixtures("plain_lru")   # watches the insert mechanics: every missed key is taken in (conftest.py)
def test_async_insert_mode_returns_default_then_converges():
    """hit rate >= hit_rate_threshold -> missed keys return the default vector now and are inserted in the
    background (docs/architecture.md:32,65-67)."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(11)
    tables = make_tables([(8000, 32)])
    keys, rows = tables[0]
    ps, cache, s = _mk("async", tables, maxcat=[1], gpucacheper=0.5, hit_rate_threshold=0.5, defaults=[7.0],
                       max_batch=4096)
    resident0 = keys[cache.query(0, keys) >= 0]
    hot = resident0[:3000]
    cold = keys[cache.query(0, keys) < 0][:300]
    q = np.concatenate([hot, cold]).astype(np.int64)
    rng.shuffle(q)
    nk = [q.size]
    out = s.lookup(q, nk).cpu().numpy()
    st = s.last_stats()
    assert st.async_insert == 1 and st.misses == cold.size
    ref_async = O.np_lookup(tables, q, nk, [7.0], resident=[resident0])
    assert np.array_equal(_bits(out), _bits(ref_async))
    cache.wait_async()
    # background insertion happened (a key whose 14-slot bucket holds nothing but keys of the current recency unit — the
    # 3,000 hot keys were just hit — stays uncached and is counted as dropped)
    c = cache.counters()
    assert (cache.query(0, cold) >= 0).mean() > 0.8
    assert int((cache.query(0, cold) >= 0).sum()) + c["dropped"] >= cold.size
    # inserting the cold keys may have evicted least-recently-used residents: whatever is resident now
    # returns its exact row, the rest the default (still async: hit rate stays above the threshold)
    resident1 = keys[cache.query(0, keys) >= 0]
    out2 = s.lookup(q, nk).cpu().numpy()
    assert s.last_stats().misses == int((~np.isin(q, resident1)).sum())
    ref2 = O.np_lookup(tables, q, nk, [7.0], resident=[resident1])
    assert np.array_equal(_bits(out2), _bits(ref2))
    assert s.last_stats().misses < cold.size


def test_lru_eviction_keeps_results_exact():
    """A cache much smaller than the working set: every call evicts; rows must stay exact."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(5)
    tables = make_tables([(20000, 64)])
    ps, cache, s = _mk("evict", tables, maxcat=[1], gpucacheper=0.01, max_batch=8192)
    co = O.COracle()
    co.add_table_arrays(*tables[0])
    for it in range(6):
        nk = [8192]
        q = _queries(rng, tables, nk, miss_frac=0.02)
        out = s.lookup(q, nk).cpu().numpy()
        ref = co.lookup(q, nk, [0.0])
        assert np.array_equal(_bits(out), _bits(ref)), it
    c = cache.counters()
    assert c["inserted"] > 0


def test_two_sessions_share_one_cache_concurrently():
    """Several lookup sessions of one model share the device cache (docs/architecture.md:20,29)."""
    import threading
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    tables = make_tables([(6000, 128), (6000, 16)])
    ps, cache, s0 = _mk("shared", tables, maxcat=[1, 1], gpucacheper=0.05, max_batch=4096)
    s1 = hps.LookupSession.create(ps, "shared", cache)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    errs = []

    def worker(sess, seed):
        rng = np.random.default_rng(seed)
        try:
            for _ in range(8):
                nk = [4096, 4096]
                q = _queries(rng, tables, nk, miss_frac=0.05)
                out = sess.lookup(q, nk).cpu().numpy()
                ref = co.lookup(q, nk, [0.0, 0.0])
                if not np.array_equal(_bits(out), _bits(ref)):
                    errs.append("mismatch")
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(s0, 100)), threading.Thread(target=worker, args=(s1, 200))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_sentinel_key_is_served_exactly():
    """INT64_MIN is the cache's empty-slot marker; as a query/table key it must still resolve exactly."""
    from oracle import hps_oracle as O
    tables = make_tables([(500, 16)])
    keys, rows = tables[0]
    keys = keys.copy()
    keys[3] = np.iinfo(np.int64).min
    tables = [(keys, rows)]
    ps, cache, s = _mk("sentinel", tables, maxcat=[1], gpucacheper=1.0, max_batch=1024)
    q = np.array([keys[3], keys[4], np.iinfo(np.int64).min, np.iinfo(np.int64).max], dtype=np.int64)
    for _ in range(2):
        out = s.lookup(q, [4]).cpu().numpy()
        ref = O.np_lookup(tables, q, [4], [0.0])
        assert np.array_equal(_bits(out), _bits(ref))


def test_refresh_embedding_cache_picks_up_new_values():
    from hugectr_backend_amd import hps
    tables = make_tables([(1000, 16)])
    keys, rows = tables[0]
    ps, cache, s = _mk("refresh", tables, maxcat=[1], gpucacheper=1.0, max_batch=2048)
    q = keys[:512].copy()
    out = s.lookup(q, [512]).cpu().numpy().reshape(512, 16)
    assert np.array_equal(_bits(out), _bits(rows[:512]))
    # new model version: same keys, new vectors -> host tier reloaded, cache refreshed
    rows2 = rows[::-1].copy()
    ps.load_table_arrays("refresh", 0, keys, rows2)
    ps.refresh_embedding_cache("refresh", 0)
    out2 = s.lookup(q, [512]).cpu().numpy().reshape(512, 16)
    assert np.array_equal(_bits(out2), _bits(rows2[:512]))


def test_refresh_drops_keys_that_left_the_parameter_server():
    """After a model update that removed keys, a cache refresh must not keep serving their stale rows."""
    from oracle import hps_oracle as O
    tables = make_tables([(1000, 16)])
    keys, rows = tables[0]
    ps, cache, s = _mk("vanish", tables, maxcat=[1], gpucacheper=1.0, defaults=[9.0], max_batch=2048)
    q = keys[:600].copy()
    out = s.lookup(q, [600]).cpu().numpy()
    assert np.array_equal(_bits(out), _bits(rows[:600].ravel()))
    keep = np.ones(keys.size, bool)
    keep[100:300] = False                                   # 200 keys disappear from the new model version
    ps.load_table_arrays("vanish", 0, keys[keep], rows[keep])
    ps.refresh_embedding_cache("vanish", 0)
    assert (cache.query(0, keys[100:300]) < 0).all()        # evicted from the device cache
    out2 = s.lookup(q, [600]).cpu().numpy()
    ref = O.np_lookup([(keys[keep], rows[keep])], q, [600], [9.0])
    assert np.array_equal(_bits(out2), _bits(ref))
    assert (out2.reshape(600, 16)[100:300] == 9.0).all()


def test_refresh_takes_only_rows_that_can_differ():
    """refresh_embedding_cache (docs/hierarchical_parameter_server.md:234-238; model_state.cpp:125-178 runs it on a timer next to
    the lookups) re-uploaded the WHOLE cache — 26.6 GB for config 2 — whether or not a row had changed.  Now: a table that was neither
    reloaded nor updated since the cache last looked costs nothing; an updated one its changed keys that are resident (each change
    replayed by two consecutive refreshes); a reloaded one a full pass; full=True and ps.json gpucache_refresh_changed_only=false
    give the reference's behaviour."""
    from oracle import hps_oracle as O
    tables = make_tables([(6000, 16), (4000, 8), (3000, 4)], seed=91)
    ps, cache, s = _mk("rf", tables, maxcat=[1, 1, 1], gpucacheper=0.5, max_batch=4096)
    T = 3
    resident = [k[cache.query(t, k) >= 0] for t, (k, _) in enumerate(tables)]
    cold = [k[cache.query(t, k) < 0] for t, (k, _) in enumerate(tables)]
    assert all(r.size > 500 for r in resident) and all(c.size > 500 for c in cold)
    # (1) nothing changed: nothing moves
    st = ps.refresh_embedding_cache("rf", 0)
    assert (st["tables"], st["tables_unchanged"], st["tables_full"]) == (T, T, 0), st
    assert st["row_bytes"] == 0 and st["keys_dumped"] == 0 and st["rows_refreshed"] == 0
    # (2) an online update of 300 resident + 200 cold keys of table 1: exactly the 300 go up, and the lookups see the new rows
    upd = np.concatenate([resident[1][:300], cold[1][:200]])
    new_rows = np.full((upd.size, 8), 5.5, np.float32)
    ps.upsert("rf", 1, upd, new_rows)
    st = ps.refresh_embedding_cache("rf", 0)
    assert (st["tables_unchanged"], st["tables_full"]) == (2, 0) and st["keys_changed"] == 500
    assert st["rows_refreshed"] == 300 and st["row_bytes"] >= 300 * 8 * 4 and st["row_bytes"] < 300 * 8 * 4 + 64
    before = cache.counters()["misses"]
    out = s.lookup(np.concatenate([tables[0][0][:0], resident[1][:300], tables[2][0][:0]]).astype(np.int64), [0, 300, 0]).cpu().numpy()
    assert np.all(out == 5.5) and cache.counters()["misses"] == before            # served from the cache, new rows
    assert (cache.query(1, cold[1][:200]) < 0).all()                              # an update is not a request: cold keys stay out
    # ... replayed once more by the next refresh (the race with a lookup that fetched the old row just before the update), then done
    st = ps.refresh_embedding_cache("rf", 0)
    assert st["rows_refreshed"] == 300 and st["tables_unchanged"] == 2
    st = ps.refresh_embedding_cache("rf", 0)
    assert st["rows_refreshed"] == 0 and st["tables_unchanged"] == 3
    # (3) a reloaded table takes a full pass, the others nothing
    k2, r2 = tables[2]
    ps.load_table_arrays("rf", 2, k2, (r2 + 1.0).astype(np.float32))
    n_res2 = int((cache.query(2, k2) >= 0).sum())
    st = ps.refresh_embedding_cache("rf", 0)
    assert (st["tables_unchanged"], st["tables_full"]) == (2, 1) and st["keys_dumped"] == n_res2 and st["rows_refreshed"] == n_res2
    q = np.concatenate([resident[2][:256]]).astype(np.int64)
    out = s.lookup(q, [0, 0, 256]).cpu().numpy()
    ref = O.np_lookup([tables[0], tables[1], (k2, (r2 + 1.0).astype(np.float32))], q, [0, 0, 256], [0.0] * 3)
    assert np.array_equal(_bits(out), _bits(ref))
    st = ps.refresh_embedding_cache("rf", 0)
    assert st["tables_unchanged"] == 3 and st["row_bytes"] == 0
    # (4) full=True: every resident row of every table, as the reference does
    n_res = [int((cache.query(t, k) >= 0).sum()) for t, (k, _) in enumerate(tables)]
    st = ps.refresh_embedding_cache("rf", 0, full=True)
    assert st["tables_full"] == 3 and st["rows_refreshed"] == sum(n_res) == st["keys_dumped"]
    assert st["row_bytes"] >= n_res[0] * 64 + n_res[1] * 32 + n_res[2] * 16
    # (5) more updates than the change log holds (4 M keys): the log is cut, the cache can no longer tell what changed — a full pass
    #     of that table, nothing for the others
    k0, r0 = tables[0]
    big = np.tile(k0, 180)[: 1_050_000]
    for _ in range(5):
        ps.upsert("rf", 0, big, np.tile(r0, (180, 1))[: big.size])
    n_res0 = int((cache.query(0, k0) >= 0).sum())
    st = ps.refresh_embedding_cache("rf", 0)
    assert (st["tables_unchanged"], st["tables_full"]) == (2, 1) and st["rows_refreshed"] == n_res0, st
    st = ps.refresh_embedding_cache("rf", 0)
    assert st["tables_unchanged"] == 3 and st["rows_refreshed"] == 0
    s.close()
    ps.close()
    # (6) ps.json gpucache_refresh_changed_only = false: every refresh is a full one
    ps, cache, s = _mk("rf2", tables[:1], maxcat=[1], gpucacheper=0.5, max_batch=4096, extra={"gpucache_refresh_changed_only": False})
    st = ps.refresh_embedding_cache("rf2", 0)
    assert st["tables_full"] == 1 and st["rows_refreshed"] == int((cache.query(0, tables[0][0]) >= 0).sum()) > 0
    s.close()


def test_lookups_are_served_exactly_while_a_full_refresh_runs():
    """Two sessions keep calling while a thread loops FULL refreshes (paced pieces, gpucache_refresh_link_share) and another one
    upserts rows: every returned row is the table's row before or after the update of its key — never torn, never another key's."""
    import threading
    tables = make_tables([(60000, 32), (40000, 16)], seed=92)
    ps, cache, s0 = _mk("rfl", tables, maxcat=[1, 1], gpucacheper=0.4, max_batch=20000)
    from hugectr_backend_amd import hps
    s1 = hps.LookupSession.create(ps, "rfl", cache)
    stop = threading.Event()
    errs, passes = [], [0]
    base = [r.copy() for _, r in tables]

    def refresher():
        try:
            while not stop.is_set():
                st = ps.refresh_embedding_cache("rfl", 0, full=True)
                assert st["tables_full"] == 2 and st["rows_refreshed"] > 0
                passes[0] += 1
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    def updater():
        rng = np.random.default_rng(3)
        try:
            v = 1
            while not stop.is_set():
                for t, (k, _) in enumerate(tables):
                    idx = rng.integers(0, k.size, 200)
                    # version v of a row: the low mantissa byte of EVERY element is v (1..255), the other 24 bits are the base row's
                    rows = ((base[t][idx].view(np.uint32) & np.uint32(0xFFFFFF00)) | np.uint32(1 + v % 255)).view(np.float32)
                    ps.upsert("rfl", t, k[idx], np.ascontiguousarray(rows))
                v += 1
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    def looker(sess, seed):
        rng = np.random.default_rng(seed)
        try:
            for it in range(60):
                nk = [15000, 9000]
                idx = [rng.integers(0, tables[t][0].size, nk[t]) for t in range(2)]
                q = np.concatenate([tables[t][0][idx[t]] for t in range(2)]).astype(np.int64)
                out = sess.lookup(q, nk).cpu().numpy()
                off = 0
                for t, D in enumerate((32, 16)):
                    got = out[off:off + nk[t] * D].reshape(nk[t], D)
                    off += nk[t] * D
                    gb, bb = got.view(np.uint32), base[t][idx[t]].view(np.uint32)
                    # one version per row: the base row itself, or every element carrying the same version byte over the base's 24 bits
                    same = (gb == bb).all(axis=1)
                    low = gb & np.uint32(0xFF)
                    versioned = ((gb & np.uint32(0xFFFFFF00)) == (bb & np.uint32(0xFFFFFF00))).all(axis=1) & (low == low[:, :1]).all(axis=1) & (low[:, 0] >= 1)
                    if not np.all(same | versioned):
                        errs.append(f"session {seed} call {it} table {t}: a torn or foreign row")
                        return
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=refresher), threading.Thread(target=updater), threading.Thread(target=looker, args=(s0, 1)),
          threading.Thread(target=looker, args=(s1, 2))]
    [t.start() for t in th]
    [t.join() for t in th[2:]]
    stop.set()
    [t.join() for t in th[:2]]
    assert not errs, errs[:3]
    assert passes[0] >= 1
    s0.close()
    s1.close()


def test_background_inserts_go_on_while_a_paced_refresh_sleeps():
    """The paced refresh sleeps most of the time (after a piece that took t it pauses for 5.7 t) — and must not hold the cache's
    inserter meanwhile: the background inserts of an async-insert model (hit_rate_threshold < 1: misses answered with the default
    vector, inserted behind the call) use the same stream and staging.  A session in async mode asks for cold keys while a thread
    loops FULL refreshes: the keys it asked for get into the cache at the rate the inserter allows, not at the rate of the refresh's
    naps."""
    import threading
    import time
    tables = make_tables([(400000, 32)], seed=44)
    keys = tables[0][0]
    ps, cache, s0 = _mk("rfasync", tables, maxcat=[1], gpucacheper=0.5, max_batch=4096, hit_rate_threshold=0.0, defaults=[7.0],
                        extra={"gpucache_load_factor": 0.5})
    cold = keys[cache.query(0, keys) < 0]
    assert cold.size > 150000
    stop = threading.Event()
    passes = [0]

    def refresher():
        while not stop.is_set():
            ps.refresh_embedding_cache("rfasync", 0, full=True)
            passes[0] += 1

    th = threading.Thread(target=refresher)
    th.start()
    asked = []
    t0 = time.time()
    i = 0
    while time.time() - t0 < 3.0:
        q = cold[i * 2000:(i + 1) * 2000].astype(np.int64)
        if q.size < 2000:
            break
        s0.lookup(q, [2000])            # async mode: defaults now, the keys go to the background inserter
        asked.append(q)
        i += 1
        time.sleep(0.01)
    stop.set()
    th.join()
    cache.wait_async()
    asked = np.concatenate(asked)
    got_in = float((cache.query(0, asked) >= 0).mean())
    # (best effort by design: a batch is dropped when num_of_worker_buffer_in_pool jobs are already waiting; here 0.75-0.80 of the
    #  keys get in with ~30 full refresh passes running meanwhile.  The refresh of THIS cache is one piece per slice, so the lock was
    #  never held for long even before the pauses released it; a config-2 cache has 160 pieces per slice.)
    print(f"async inserts under a paced full refresh: {got_in:.3f} of {asked.size} keys got in, {passes[0]} refresh passes")
    assert got_in > 0.5, (got_in, len(asked), passes[0])
    s0.close()


@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_per_call_switches_on_their_threshold_two_sessions_in_opposite_phase(direct):
    """The per-call switches "keys_by_kernel" 2 and "probe_in_lane" 2 (and "interact_mode" 2, tests/test_gpu_dense.py) steer by
    'this session's calls miss much'.  Round 5 compared the last call's missed rows with side_scatter_mb on every call: traffic on
    the bound flipped the arrangement call by call.  Two sessions of one cache whose consecutive calls straddle the bound — missed
    rows alternately 1.15 x and 0.65 x side_scatter_mb — in OPPOSITE phase for 240 calls each: rows exact, at most one call in 50 above three
    times the median (the boxes' lone 4-ms scheduler-tick calls are not a mode), and the mode (a switch with hysteresis and a dwell of 8 calls, hps_lookup_stats_t::mode_flips) changes at most once
    per 8 calls."""
    import threading
    import torch
    from hugectr_backend_amd import hps
    T, R, D = 2, 200000, 64
    tables = make_tables([(R, D)] * T, seed=61)
    ps, cache, s0 = _mk(f"thr{int(direct)}", tables, maxcat=[1] * T, gpucacheper=1.0, max_batch=90000, defaults=[2.5, -1.0],
                        extra={"ps_direct_access": direct})
    s1 = hps.LookupSession.create(ps, f"thr{int(direct)}", cache)
    bound_rows = (1 << 20) // (D * 4)                      # side_scatter_mb = 1: 4,096 rows of 256 bytes
    nk = [85000, 85000]                                    # (a "big" request: its keys are staged in pieces by the pool)
    rows_d = [torch.from_numpy(r).cuda() for _, r in tables]
    dflt = [torch.full((D,), v, device="cuda") for v in (2.5, -1.0)]
    # (a few per cent of the rows found their bucket full at warm-up: the requests draw from what IS resident, so that the keys that
    #  exist nowhere are the only misses)
    res_idx = [np.nonzero(cache.query(t, tables[t][0]) >= 0)[0] for t in range(T)]
    calls = 240
    errs, lat, final = [], [[], []], [None, None]

    # every request and the rows it must return, made up front (20 GB of expected rows in HBM): inside the loop there is nothing but
    # the lookup and one comparison kernel — tensors created per call cost allocator and copy calls that hold the runtime's lock
    # under the OTHER session's lookup (0.5-ms calls in a first version of this test)
    def prepare(i):
        rng = np.random.default_rng(100 + i)
        reqs = []
        for c in range(calls):
            miss = int(bound_rows * (1.15 if (c + i) % 2 == 0 else 0.65))
            parts, exp = [], []
            for t in range(T):
                idx = res_idx[t][rng.integers(0, res_idx[t].size, nk[t])]
                q = tables[t][0][idx].astype(np.int64)
                m = miss // T
                pos = rng.choice(nk[t], m, replace=False)
                q[pos] = -10 - (np.arange(m, dtype=np.int64) + (c * 4 + t) * 100000)      # distinct keys that exist nowhere: m unique misses
                e_ = rows_d[t][torch.from_numpy(idx).cuda()]
                e_[torch.from_numpy(pos).cuda()] = dflt[t]
                parts.append(q)
                exp.append(e_.reshape(-1))
            reqs.append((np.concatenate(parts), torch.cat(exp), (miss // T) * T))
        return reqs

    prepared = [prepare(0), prepare(1)]
    torch.cuda.synchronize()

    # Two passes over the same requests: one that checks every row (torch.equal per call), one that only times the calls.  A
    # comparison inside the timed loop showed up in the OTHER session's calls (its device-to-host read-back holds the runtime's
    # lock: one 3.9-ms call in 480) — nothing of the library's doing, and not what this test is about.
    def work(i, sess, verify):
        try:
            sess.set_option("side_scatter_mb", 1)
            out = torch.empty(sum(nk) * D, dtype=torch.float32, device="cuda")
            for c, (q, exp, um) in enumerate(prepared[i]):
                sess.lookup(q, nk, out=out)
                st = sess.last_stats()
                assert st.unique_misses == um, (st.unique_misses, um)
                if verify:
                    if not torch.equal(out, exp):
                        errs.append((i, c))
                        return
                else:
                    lat[i].append(float(st.phase_ms[3]) + float(st.key_stage_ms))              # the call as the engine timed it
            final[i] = sess.last_stats()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    def throttled():      # the container's CPU quota (cgroup cpu.max): a throttled period freezes every thread of the process for milliseconds
        try:
            return int([l.split()[1] for l in open("/sys/fs/cgroup/cpu.stat") if l.startswith("nr_throttled")][0])
        except Exception:  # noqa: BLE001
            return 0

    passes = 0
    for verify in (True, False, False, False):
        if not verify:
            lat[0].clear(); lat[1].clear()
        thr0 = throttled()
        th = [threading.Thread(target=work, args=(0, s0, verify)), threading.Thread(target=work, args=(1, s1, verify))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs[:3]
        passes += 1
        # A timed pass during which the process was frozen by its CPU quota says nothing about the library — and neither does one
        # with a single 4-ms call in it: the boxes show such calls now and then whatever runs (a thread of the HIP runtime that
        # releases work queued behind a cross-stream event loses its CPU for a scheduler tick).  A bad MODE — what this
        # test is after — repeats; a tick does not: the pass is run again, three times at most, and the last one is judged.
        if not verify and throttled() == thr0 and all(max(l[5:]) < 3 * float(np.median(l)) for l in lat):
            break
    calls *= passes
    # Judged: a MODE, not a tick.  After up to three timed passes the last one may still hold a lone 4-ms call (some boxes of the pool
    # show them in every leg, with or without this library's switches: profiles/round6/r6z's hit_999 leg has a 7-ms call with nothing
    # switching at all) — what a bad arrangement would do is slow MANY calls: at most one call in 50 may exceed 3 x the median, and
    # the 97th percentile must not.
    for i in range(2):
        a = np.asarray(lat[i][5:])
        med = float(np.median(a))
        assert float(np.percentile(a, 97)) < 3 * med and float((a > 3 * med).mean()) <= 0.02, (i, med, float(a.max()), float((a > 3 * med).mean()))
        # 1.15 x the bound switches up, 0.65 x (below three quarters) down, each change then holds for 8 calls
        assert 2 <= final[i].mode_flips <= calls // 8, final[i].mode_flips
    s0.close()
    s1.close()


def test_call_counter_wrap(monkeypatch):
    """The recency unit travels in 24 bits of a call's time token and wraps freely (the stamps in the bucket lines are the
    unit modulo 255 and repeat one value at the wrap); results stay exact and insertion keeps working across it.  Call clock,
    one call per unit, started four calls before the wrap."""
    from oracle import hps_oracle as O
    monkeypatch.setenv("HPS_LRU_AGE_SHIFT", "0")
    monkeypatch.setenv("HPS_TEST_EPOCH_START", str(2**24 - 4))
    rng = np.random.default_rng(8)
    tables = make_tables([(20000, 32)])
    ps, cache, s = _mk("renorm", tables, maxcat=[1], gpucacheper=0.02, max_batch=4096)
    monkeypatch.delenv("HPS_TEST_EPOCH_START")
    co = O.COracle()
    co.add_table_arrays(*tables[0])
    inserted_before = cache.counters()["inserted"]
    for it in range(10):                                    # the counter wraps at the 4th call
        q = _queries(rng, tables, [4096], miss_frac=0.02)
        out = s.lookup(q, [4096]).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(co.lookup(q, [4096], [0.0]))), it
    assert cache.counters()["inserted"] > inserted_before  # eviction/insertion keeps working after the wrap


@pytest.mark.usefixtures("plain_lru")   # watches the insert mechanics: every missed key is taken in (conftest.py)
@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_insertion_policy_is_decided_per_table(direct):
    """hit_rate_threshold between two tables' hit rates: the hot table answers in async-insert mode (defaults for its
    misses, background insertion), the cold table synchronously (exact rows, inserted before return) — in ONE call
    (the reference decides inside its per-table loop)."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(77)
    tables = make_tables([(8000, 32), (8000, 64), (500, 16)])
    ps, cache, s = _mk("policy_d" if direct else "policy_h", tables, maxcat=[1, 1, 1], gpucacheper=0.5, hit_rate_threshold=0.8,
                       defaults=[7.0, -3.0, 0.5], max_batch=4096, extra={"ps_direct_access": direct})
    resident = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
    cold = [tk[cache.query(t, tk) < 0] for t, (tk, _) in enumerate(tables)]
    n = 3000
    q0 = np.concatenate([rng.choice(resident[0], n - 150), rng.choice(cold[0], 150)])      # 95 % hit -> async
    q1 = np.concatenate([rng.choice(resident[1], n // 2), rng.choice(cold[1], n // 2)])    # 50 % hit -> sync
    q2 = rng.choice(resident[2], 200)                                                       # all hit
    for a in (q0, q1):
        rng.shuffle(a)
    nk = [q0.size, q1.size, q2.size]
    q = np.concatenate([q0, q1, q2]).astype(np.int64)
    modes = O.np_insert_modes(q, nk, resident, 0.8)
    assert modes == [True, False, False]
    out = s.lookup(q, nk).cpu().numpy()
    st = s.last_stats()
    assert st.async_insert == 1
    assert st.misses == 150 + n // 2
    ref = O.np_lookup(tables, q, nk, [7.0, -3.0, 0.5], resident=[resident[0], None, None])
    assert np.array_equal(_bits(out), _bits(ref))
    # the synchronous table's missed keys are resident when the call returns; the async table's after the background job
    # (a few may have been dropped from over-full buckets, as in every insert)
    assert (cache.query(1, cold[1]) >= 0).sum() >= 0.8 * np.unique(q1[np.isin(q1, cold[1])]).size
    newly0 = np.unique(q0[np.isin(q0, cold[0])])
    cache.wait_async()
    assert (cache.query(0, newly0) >= 0).mean() > 0.8
    # second call, same keys: everything resident now -> no async part, exact rows everywhere
    res2 = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
    modes2 = O.np_insert_modes(q, nk, res2, 0.8)
    out2 = s.lookup(q, nk).cpu().numpy()
    ref2 = O.np_lookup(tables, q, nk, [7.0, -3.0, 0.5], resident=[res2[t] if modes2[t] else None for t in range(3)])
    assert np.array_equal(_bits(out2), _bits(ref2))


@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_cold_batch_larger_than_one_staging_chunk(direct):
    """600 K distinct missed rows of 512 B = 307 MB: more than the 256-MB staging chunk of the host-gather path, so its
    miss loop runs twice (the device-driven tier stages the whole call at once); rows exact either way."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    R, D, n = 700_000, 128, 600_000
    keys = np.arange(R, dtype=np.int64)
    rows = O.c_synth_rows(O.SEED, 0, 0, R, D)
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
           "models": [{"model": "cold", "sparse_files": ["x"], "num_of_worker_buffer_in_pool": 1,
                       "embedding_vecsize_per_table": [D], "maxnum_catfeature_query_per_table_per_sample": [1],
                       "default_value_for_each_table": [0.0], "deployed_device_list": [0], "max_batch_size": n,
                       "gpucache": True, "gpucacheper": 0.01, "hit_rate_threshold": 1.0, "ps_direct_access": direct}]}
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    ps.load_table_arrays("cold", 0, keys, rows)
    ps.create_embedding_cache_per_model("cold")
    s = hps.LookupSession.create(ps, "cold", ps.get_embedding_cache("cold", 0))
    q = np.random.default_rng(4).permutation(R)[:n].astype(np.int64)
    out = s.lookup(q, [n]).cpu().numpy().reshape(n, D)
    assert s.last_stats().unique_misses > 590_000
    assert np.array_equal(out.view(np.uint32), rows[q].view(np.uint32))


@pytest.mark.parametrize("direct", [False], ids=["host_gather"])   # the device-driven tier keeps the fused kernel
@pytest.mark.parametrize("dims", [[128, 128, 128], [16, 128, 1, 8]], ids=["all_128", "mixed_widths"])
def test_split_probe_returns_the_same_rows(dims, direct):
    """Option split_probe (host-gather tier): the miss counts are read back right behind the probe and the hit rows are
    moved by K_G while the misses are fetched.  Same rows and counts as the un-split order, through ragged requests with
    duplicates, absent keys, empty tables, an all-hit call and the async-insert policy."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(len(dims))
    tables = make_tables([(30000 + 1000 * t, d) for t, d in enumerate(dims)])
    defaults = [0.25 * (t + 1) for t in range(len(dims))]
    ps, cache, s = _mk(f"split{len(dims)}{int(direct)}", tables, maxcat=[1] * len(dims), defaults=defaults, gpucacheper=0.1,
                       max_batch=20000, extra={"ps_direct_access": direct})
    s.set_option("split_probe", 1)
    s.set_option("timing", 1)
    saw_split = 0
    for it in range(14):
        nk = [int(rng.integers(0, 15000)) for _ in dims]
        if it == 5:
            nk[1] = 0
        q = _queries(rng, tables, nk, miss_frac=0.1)
        if it in (8, 9):   # resident keys only -> no misses -> the next call goes back to the fused kernel
            q = np.concatenate([rng.choice(tk[cache.query(t, tk) >= 0], n) for t, ((tk, _), n) in enumerate(zip(tables, nk))]).astype(np.int64)
        misses = uniq = off = 0
        for t, n in enumerate(nk):
            qt = q[off:off + n]
            off += n
            m = qt[cache.query(t, qt) < 0] if n else qt
            misses += m.size
            uniq += np.unique(m).size
        out = s.lookup(q, nk).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, nk, defaults))), it
        st = s.last_stats()
        assert (st.misses, st.unique_misses) == (misses, uniq), it
        saw_split += st.hit_gather_ms > 0
    assert saw_split >= 8
    # async-insert policy on top of a split call: defaults for the missed keys of the async tables
    s.set_option("hit_rate_threshold_permille", 500)
    nk = [8000] * len(dims)
    q = _queries(rng, tables, nk, miss_frac=0.05)
    res = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
    modes = O.np_insert_modes(q, nk, res, 0.5)
    out = s.lookup(q, nk).cpu().numpy()
    ref = O.np_lookup(tables, q, nk, defaults, resident=[res[t] if modes[t] else None for t in range(len(dims))])
    assert np.array_equal(_bits(out), _bits(ref))


def test_policy_hit_rate_is_over_unique_keys():
    """SURVEY.md App. C4 / docs/hierarchical_parameter_server.md:69: the hit rate behind the sync/async decision is
    1 - unique misses / unique keys of the table in this call.  Two skewed tables on which the keys-as-sent reading
    decides the opposite way; the product must follow the oracle (np_insert_modes) and report the unique-key count."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(404)
    tables = make_tables([(6000, 32), (6000, 16)])
    ps, cache, s = _mk("uniqpol", tables, maxcat=[1, 1], gpucacheper=0.5, hit_rate_threshold=0.8, defaults=[5.0, -2.0],
                       max_batch=8192)
    resident = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
    cold = [tk[cache.query(t, tk) < 0] for t, (tk, _) in enumerate(tables)]
    # table 0: one resident key sent 7,000 times + 300 distinct cold keys: 96 % hit as sent, 1/301 over unique keys -> sync
    q0 = np.concatenate([np.full(7000, resident[0][5]), cold[0][:300]])
    # table 1: 2,000 distinct resident keys + one cold key sent 5,000 times: 29 % hit as sent, 2000/2001 unique -> async
    q1 = np.concatenate([resident[1][:2000], np.full(5000, cold[1][7])])
    rng.shuffle(q0)
    rng.shuffle(q1)
    nk = [q0.size, q1.size]
    q = np.concatenate([q0, q1]).astype(np.int64)
    modes = O.np_insert_modes(q, nk, resident, 0.8)
    assert modes == [False, True]
    assert O.np_insert_modes_keys_as_sent(q, nk, resident, 0.8) == [True, False]   # this test fails on that definition
    out = s.lookup(q, nk).cpu().numpy()
    ref = O.np_lookup(tables, q, nk, [5.0, -2.0], resident=[None, resident[1]])
    assert np.array_equal(_bits(out), _bits(ref))
    wrong = O.np_lookup(tables, q, nk, [5.0, -2.0], resident=[resident[0], None])
    assert not np.array_equal(_bits(out), _bits(wrong))
    st = s.last_stats()
    uc = O.np_unique_counts(q, nk, resident)
    assert st.async_insert == 1
    assert st.unique_keys == sum(u for u, _ in uc) == 301 + 2001
    assert st.unique_misses == sum(m for _, m in uc) == 301
    assert st.misses == 300 + 5000


def test_unique_hit_count_over_a_cache_of_several_bitmap_parts():
    """K_H (hps_unique_hits_kernel) counts a table's distinct hit slots in LDS bitmaps of 2^20 slots each: a cache of 3.3 M and
    one of 1.2 M slots (four and two workgroups), a third table with no key in the call, keys repeated across the whole range,
    key ranges that do not start on a 16-byte boundary of the slot array.  The count must be the oracle's."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(2026)
    tables = make_tables([(2_500_000, 1), (900_000, 2), (1000, 4)])
    ps, cache, s = _mk("khparts", tables, maxcat=[3, 2, 1], gpucacheper=1.0, hit_rate_threshold=0.5, defaults=[1.0, 2.0, 3.0],
                       max_batch=70_000)
    resident = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
    assert resident[0].size > 2_400_000 and resident[1].size > 850_000
    for it, nk in enumerate([[200_001, 130_003, 0], [77_777, 5, 0], [3, 139_999, 0]]):
        parts = []
        for t, n in enumerate(nk):
            if n == 0:
                parts.append(np.zeros(0, np.int64)); continue
            hot = rng.choice(resident[t], size=max(1, n // 3), replace=False)      # distinct slots all over the table
            q = rng.choice(hot, size=n, replace=True)                               # ... each several times
            absent = rng.random(n) < 0.1
            parts.append(np.where(absent, -1 - rng.integers(0, 1 << 40, n), q).astype(np.int64))
        q = np.concatenate(parts)
        uc = O.np_unique_counts(q, nk, resident)
        out = s.lookup(q, nk).cpu().numpy()
        st = s.last_stats()
        assert st.unique_keys == sum(u for u, _ in uc), (it, st.unique_keys, uc)
        assert st.unique_misses == sum(m for _, m in uc), it
        modes = O.np_insert_modes(q, nk, resident, 0.5)
        ref = O.np_lookup(tables, q, nk, [1.0, 2.0, 3.0], resident=[resident[t] if modes[t] else None for t in range(3)])
        assert np.array_equal(_bits(out), _bits(ref)), it
        cache.wait_async()


@pytest.mark.parametrize("variant", [1002, 1102, -1002])
@pytest.mark.parametrize("xcd_walk", [0, 1])
def test_probe_variants_and_gather_walks_agree(variant, xcd_walk):
    """Every instantiation of the probe kernel that ships (with the tile-local input dedup — unique misses in its tail or, -1002,
    in a launch of their own — and without it) and both chunk walks of the gather kernel return the same rows and counts: ragged tables that end inside a
    tile, tiles full of one key, keys absent everywhere, the sentinel key, an empty table."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(abs(variant) * 2 + xcd_walk)
    tables = make_tables([(9000, 128), (5000, 128), (300, 4), (4000, 128)])
    defaults = [0.5, 1.5, 2.5, 3.5]
    ps, cache, s = _mk(f"var{variant}_{xcd_walk}", tables, maxcat=[2, 1, 1, 1], defaults=defaults, gpucacheper=0.3, max_batch=8192,
                       hit_rate_threshold=0.9)
    s.set_option("probe_variant", abs(variant))
    if variant < 0:
        s.set_option("fused_unique", 0)
    s.set_option("xcd_walk", xcd_walk)
    for it in range(6):
        nk = [int(rng.integers(1, 12000)), int(rng.integers(0, 8000)), int(rng.integers(0, 2100)), 0 if it == 2 else int(rng.integers(1, 3000))]
        q = _queries(rng, tables, nk, miss_frac=0.03)
        if it == 1:   # a whole tile (and more) of one key, resident or not
            q[:2500] = tables[0][0][11]
            q[nk[0]:nk[0] + min(nk[1], 1500)] = -77
        if it == 3:
            q[::97] = np.int64(-2**63)   # the engine's empty-slot marker is a legal query
        res = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
        modes = O.np_insert_modes(q, nk, res, 0.9)
        uc = O.np_unique_counts(q, nk, res)
        out = s.lookup(q, nk).cpu().numpy()
        ref = O.np_lookup(tables, q, nk, defaults, resident=[res[t] if modes[t] else None for t in range(4)])
        assert np.array_equal(_bits(out), _bits(ref)), it
        st = s.last_stats()
        assert st.unique_misses == sum(m for _, m in uc), it
        assert st.unique_keys == sum(u for u, _ in uc), it
        cache.wait_async()


def test_host_keys_staged_in_pieces_and_pinned_in_place():
    """hps_session_lookup's three ways of getting host keys to the device — small request (one staging copy), large
    pageable request (pieces staged and uploaded by the serving pool), flat page-locked array (DMA in place, checked
    with and without the pointer check) — return the same rows."""
    import torch
    from oracle import hps_oracle as O
    rng = np.random.default_rng(8)
    tables = make_tables([(50000, 16), (50000, 8), (1000, 4)])
    ps, cache, s = _mk("hostkeys", tables, maxcat=[1, 1, 1], gpucacheper=0.3, max_batch=700_000)
    for nk in ([1000, 0, 50], [650_000, 640_123, 999]):
        q = _queries(rng, tables, nk, miss_frac=0.02)
        ref = O.np_lookup(tables, q, nk, [0.0] * 3)
        out = s.lookup(q, nk).cpu().numpy()                       # pageable numpy memory
        assert np.array_equal(_bits(out), _bits(ref))
        assert s.last_stats().key_stage_ms > 0
        pinned = torch.from_numpy(q).pin_memory()
        for check in (1, 0):
            s.set_option("keys_pinned_check", check)
            out = s.lookup(pinned.numpy(), nk).cpu().numpy()
            assert np.array_equal(_bits(out), _bits(ref))
            if check == 1 and sum(nk) > 200_000:
                # the DMA-in-place branch was taken: page-locked keys are never read by host threads (an order of magnitude
                # slower than pageable memory: round 2's 87-ms calls) — they cross PCIe as they are, 8 bytes each, and the
                # host side of the staging is the pointer check plus one enqueue
                st = s.last_stats()
                assert st.key_bytes == 8 and st.keys_narrowed == 0
                assert st.key_stage_ms < 1.0, st.key_stage_ms
        s.set_option("keys_pinned_check", 1)
        # per-table pointers that are NOT one flat array (each table from its own allocation)
        parts, off = [], 0
        for n in nk:
            parts.append(np.array(q[off:off + n], copy=True))
            off += n
        outp = torch.empty(ref.size, dtype=torch.float32, device="cuda")
        koff = 0
        vptrs = []
        for t, n in enumerate(nk):
            vptrs.append(outp.data_ptr() + 4 * koff)
            koff += n * tables[t][1].shape[1]
        s.lookup_ptrs([p.ctypes.data if p.size else 0 for p in parts], vptrs, nk)
        assert np.array_equal(_bits(outp.cpu().numpy()), _bits(ref))


def test_pageable_keys_cross_pcie_as_uint32_when_they_fit():
    """Staging narrows a request's keys to 32 bits when every key fits (checked while copying); one wide or negative key
    anywhere in the request makes that call use the 8-byte copy; rows are exact either way."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(32)
    tables = make_tables([(60000, 16), (60000, 8)])
    big = tables[1][0].copy()
    big[:100] += 1 << 40            # a table that also holds keys beyond 32 bits
    tables[1] = (big, tables[1][1])
    ps, cache, s = _mk("narrow", tables, maxcat=[1, 1], gpucacheper=0.3, max_batch=400_000)
    nk = [300_000, 250_000]
    q = np.concatenate([rng.choice(tables[0][0], nk[0]), rng.choice(big[100:], nk[1])]).astype(np.int64)
    q[rng.integers(0, q.size, 5000)] = (1 << 32) - 1 - rng.integers(0, 1000, 5000)      # absent, but 32-bit
    out = s.lookup(q, nk).cpu().numpy()
    assert s.last_stats().keys_narrowed == 1 and s.last_stats().key_bytes == 4   # 3-byte packing tried first, keys of 32 bits seen
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, nk, [0.0, 0.0])))
    q2 = q.copy()
    q2[nk[0] + 123_456] = big[7]      # one key of 41 bits in the last staging group
    out = s.lookup(q2, nk).cpu().numpy()
    assert s.last_stats().keys_narrowed == 0
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q2, nk, [0.0, 0.0])))
    out = s.lookup(q, nk).cpu().numpy()          # the session backs off for a while after a wide key
    assert s.last_stats().keys_narrowed == 0
    s.set_option("narrow_keys", 1)                # re-arms it
    out = s.lookup(q, nk).cpu().numpy()
    assert s.last_stats().keys_narrowed == 1
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, nk, [0.0, 0.0])))
    q3 = q.copy()
    q3[5] = -9
    out = s.lookup(q3, nk).cpu().numpy()
    assert s.last_stats().keys_narrowed == 0
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q3, nk, [0.0, 0.0])))


def test_copy_engines_are_woken_once_per_device():
    """Cache creation sends one tiny copy through every SDMA engine (copy_engines.h); the C ABI reports what it did and
    a second call is a table look-up."""
    import time
    from hugectr_backend_amd import hps
    n, report = hps.wake_copy_engines(0)
    assert n >= 2 and "host->device" in report, report
    t0 = time.perf_counter()
    assert hps.wake_copy_engines(0) == (n, report)
    assert time.perf_counter() - t0 < 0.01


@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_control_words_by_kernels_or_by_copies_same_rows(direct, monkeypatch):
    """The call block, the miss counts, a small request's keys and a small miss chunk travel by kernels that touch page-locked
    memory (default) or by hipMemcpyAsync (HPS_ZC_CONTROL=0, read when a session is created).  Both arrangements, request
    sizes on either side of the small-request limit (131,072 keys) and miss volumes on either side of the in-place limit
    (256 KB of rows), pageable and page-locked keys, three calls each: exact rows."""
    import torch
    from oracle import hps_oracle as O
    tables = make_tables([(80000, 16), (80000, 32), (2000, 4)])
    for zc in ("1", "0"):
        monkeypatch.setenv("HPS_ZC_CONTROL", zc)
        rng = np.random.default_rng(77)
        ps, cache, s = _mk(f"zc{zc}_{int(direct)}", tables, maxcat=[1, 1, 1], gpucacheper=0.25, max_batch=200_000,
                           extra={"ps_direct_access": direct})
        for nk, miss in (([900, 1100, 40], 0.02), ([60_000, 60_000, 900], 0.002), ([60_000, 60_000, 900], 0.3),
                         ([150_000, 140_000, 1500], 0.001), ([150_000, 140_000, 1500], 0.2)):
            for rep in range(3):
                q = _queries(rng, tables, nk, miss_frac=miss)
                ref = O.np_lookup(tables, q, nk, [0.0] * 3)
                keys = torch.from_numpy(q).pin_memory().numpy() if rep == 1 else q
                out = s.lookup(keys, nk).cpu().numpy()
                assert np.array_equal(_bits(out), _bits(ref)), (zc, nk, miss, rep)
        s.close()


def test_pageable_keys_below_2_to_24_cross_pcie_at_three_bytes():
    """Keys that all fit 24 bits are packed at 3 bytes each while staging; one key of 25..32 bits anywhere makes that call use
    uint32 (and the session stops trying the packing for a while); exact rows either way, also when a table's slice starts
    at an odd byte offset of the packed array."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(24)
    tables = make_tables([(70000, 16), (70000, 8), (500, 4)])
    keys1 = tables[1][0].copy()
    keys1[:50] += 1 << 27            # a few keys of table 1 need 28 bits
    tables[1] = (keys1, tables[1][1])
    ps, cache, s = _mk("pack24", tables, maxcat=[1, 1, 1], gpucacheper=0.3, max_batch=400_000)
    nk = [200_001, 180_003, 333]
    small = [tables[0][0], keys1[50:], tables[2][0]]
    q = np.concatenate([rng.choice(k, n) for k, n in zip(small, nk)]).astype(np.int64)
    q[rng.integers(0, q.size, 3000)] = (1 << 24) - 1 - rng.integers(0, 500, 3000)       # absent, 24 bits
    for rep in range(2):
        out = s.lookup(q, nk).cpu().numpy()
        st = s.last_stats()
        assert (st.keys_narrowed, st.key_bytes) == (1, 3)
        assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, nk, [0.0] * 3)))
    q2 = q.copy()
    q2[nk[0] + 150_000] = keys1[3]       # 28 bits, in a late staging group of table 1
    out = s.lookup(q2, nk).cpu().numpy()
    assert s.last_stats().key_bytes == 4
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q2, nk, [0.0] * 3)))
    out = s.lookup(q, nk).cpu().numpy()                # packing is not tried again right away
    assert s.last_stats().key_bytes == 4
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, nk, [0.0] * 3)))
    s.set_option("narrow_keys", 1)                     # re-arms it
    out = s.lookup(q, nk).cpu().numpy()
    assert s.last_stats().key_bytes == 3
    q3 = q.copy()
    q3[7] = 1 << 40
    out = s.lookup(q3, nk).cpu().numpy()
    assert s.last_stats().key_bytes == 8 and s.last_stats().keys_narrowed == 0
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q3, nk, [0.0] * 3)))


@pytest.mark.parametrize("fused", [1, 0])
def test_call_wide_unique_misses_with_the_same_missed_keys_in_every_tile(fused):
    """The call-wide dedup of missed keys — in the probe kernel's tail (default: a loser reads the winner's key, which another
    workgroup of the same launch wrote) or in the separate hps_miss_unique launch: requests whose tiles all miss the SAME few
    keys (absent ones and non-resident ones), so that nearly every missed representative is a loser of some other tile's
    entry.  Many calls on one session (set entries of earlier calls are stale, never cleared); the unique-miss count must be
    exactly the number of distinct missed keys and the rows exact, call after call."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(1234 + fused)
    tables = make_tables([(30000, 32), (30000, 8), (30000, 128)])
    ps, cache, s = _mk(f"fuq{fused}", tables, maxcat=[1, 1, 1], gpucacheper=0.05, hit_rate_threshold=1.0, defaults=[0.25, 0.5, 0.75],
                       max_batch=40000)
    s.set_option("fused_unique", fused)
    for it in range(60):
        nk, parts = [], []
        for t, (keys, _) in enumerate(tables):
            n = int(rng.integers(9000, 40000))                    # 9 .. 40 tiles of this table
            pool_cold = rng.choice(keys, size=int(rng.integers(1, 120)), replace=False)      # mostly non-resident (cache = 5 %)
            pool_absent = -7 - rng.integers(0, 1 << 45, size=int(rng.integers(1, 60)))      # in no table
            pool = np.concatenate([pool_cold, pool_absent])
            q = rng.choice(pool, size=n, replace=True)
            nk.append(n)
            parts.append(q.astype(np.int64))
        q = np.concatenate(parts)
        resident = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
        uc = O.np_unique_counts(q, nk, resident)
        out = s.lookup(q, nk).cpu().numpy()
        ref = O.np_lookup(tables, q, nk, [0.25, 0.5, 0.75])
        assert np.array_equal(_bits(out), _bits(ref)), it
        st = s.last_stats()
        assert st.unique_misses == sum(m for _, m in uc), (it, st.unique_misses, uc)
    c = cache.counters()
    assert c["inserted"] + c["refreshed"] + c["dropped"] > 0
    s.close()


def test_fused_unique_tail_against_the_separate_kernel_on_full_size_duplicate_heavy_batches():
    """Advisor finding of round 3 (kernels.hip, the probe kernel's tail): a loser of the call-wide set reads the WINNER's key,
    written moments earlier by another workgroup of the same launch — possibly on another XCD, whose L2 is not coherent with
    the reader's.  The code relies on what device-scope atomics are on gfx950 (sc1 stores write through the XCD's L2 and are
    counted by vmcnt until acknowledged at the device's coherence point; sc1 loads are not served from a stale L2 line): a
    stale read would either miss a duplicate (a key fetched and inserted twice in one launch: unique count too high) or match
    a wrong representative (a wrong row).  Stress: the headline's shape — 26 tables x 65,536 keys = 1,664 tiles, every XCD
    busy — with ALL 64 tiles of a table drawing their misses from the same 400 keys, many calls on one session (stale set
    entries of earlier calls), the tail against the separate hps_miss_unique launch: same unique counts, to the key, and
    exact rows."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    rng = np.random.default_rng(4242)
    T, R, D, B = 26, 40000, 8, 65536
    tables = make_tables([(R, D)] * T)
    ps, cache, s1 = _mk("fuqbig", tables, maxcat=[1] * T, gpucacheper=0.05, hit_rate_threshold=1.0, defaults=[0.5] * T, max_batch=B,
                        extra={"gpucache_admission": False})
    s0 = hps.LookupSession.create(ps, "fuqbig", cache)
    s1.set_option("fused_unique", 1)
    s0.set_option("fused_unique", 0)
    nk = [B] * T
    for it in range(24):
        resident = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
        parts = []
        for t, (keys, _) in enumerate(tables):
            cold = np.setdiff1d(keys, resident[t])
            pool = np.concatenate([rng.choice(cold, 300, replace=False), -7 - rng.integers(0, 1 << 45, size=100)])
            q = rng.choice(resident[t], B)
            miss = rng.random(B) < 0.3
            q[miss] = rng.choice(pool, int(miss.sum()))
            parts.append(q.astype(np.int64))
        q = np.concatenate(parts)
        uc = O.np_unique_counts(q, nk, resident)
        want_unique = sum(m for _, m in uc)
        ref = O.np_lookup(tables, q, nk, [0.5] * T)
        # the tail first (it meets the keys as misses); the separate kernel afterwards sees what the first call could not
        # insert (absent keys, dropped keys) — so it is checked on a fresh draw of its own every other round
        sess = s1 if it % 2 == 0 else s0
        out = sess.lookup(q, nk).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(ref)), it
        assert sess.last_stats().unique_misses == want_unique, (it, sess.last_stats().unique_misses, want_unique)
    s0.close()
    s1.close()


def test_admission_rule_keeps_recently_hit_keys_and_lets_new_keys_in_once_they_aged(monkeypatch):
    """Default insertion policy (kernels.hip, hps_cache_insert_kernel): a new key's nominal age is the insert age, so it never
    takes a slot that was hit more recently than that — its row is served exactly, it just stays out of the cache; once
    the bucket's keys have gone unhit for longer than the insert age, new keys replace them.  Here: recency unit = 1 call,
    insert age 4 calls, no bypass (HPS_LRU_ADMIT=15: one new key in 32,768 would be let through regardless)."""
    from oracle import hps_oracle as O
    monkeypatch.setenv("HPS_LRU_AGE_SHIFT", "0")
    monkeypatch.setenv("HPS_LRU_INSERT_AGE", "4")
    monkeypatch.setenv("HPS_LRU_ADMIT", "15")
    tables = make_tables([(8000, 32)])
    keys, rows = tables[0]
    ps, cache, s = _mk("admit", tables, maxcat=[1], gpucacheper=0.5, defaults=[0.0], max_batch=8192)
    resident0 = keys[cache.query(0, keys) >= 0]
    cold = keys[cache.query(0, keys) < 0]
    s.lookup(resident0.astype(np.int64), [resident0.size])                                  # every resident key is hit ...
    q = cold[:400].astype(np.int64)                                                         # ... one call before 400 new keys arrive
    out = s.lookup(q, [q.size]).cpu().numpy()
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, [q.size], [0.0])))      # rows exact whatever is admitted
    c1 = cache.counters()
    assert (cache.query(0, resident0) >= 0).sum() >= resident0.size - 1                    # nobody hit a call ago was given up
    in1 = int((cache.query(0, cold[:400]) >= 0).sum())
    assert 0 < in1 < 400 and c1["dropped"] >= 400 - in1 - 1                                # free slots taken, full buckets refuse
    for _ in range(6):                                                                      # six calls go by; the residents are not asked for
        s.lookup(resident0[:1].astype(np.int64), [1])
    q2 = cold[400:800].astype(np.int64)
    out2 = s.lookup(q2, [q2.size]).cpu().numpy()
    assert np.array_equal(_bits(out2), _bits(O.np_lookup(tables, q2, [q2.size], [0.0])))
    assert (cache.query(0, q2) >= 0).mean() > 0.97                                         # older than the insert age: replaced
    # plain LRU insertion for comparison (HPS_LRU_ADMIT=0): the same two calls evict keys that were hit a call ago
    monkeypatch.setenv("HPS_LRU_ADMIT", "0")
    ps0, cache0, s0 = _mk("admit0", tables, maxcat=[1], gpucacheper=0.5, defaults=[0.0], max_batch=8192)
    r0 = keys[cache0.query(0, keys) >= 0]
    c0 = keys[cache0.query(0, keys) < 0]
    s0.lookup(r0.astype(np.int64), [r0.size])
    q = c0[:400].astype(np.int64)
    out = s0.lookup(q, [q.size]).cpu().numpy()
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, [q.size], [0.0])))
    assert (cache0.query(0, q) >= 0).mean() > 0.97 and (cache0.query(0, r0) >= 0).sum() < r0.size - 20
    # the same through ps.json ("gpucache_admission": false) instead of the environment
    monkeypatch.delenv("HPS_LRU_ADMIT")
    ps1, cache1, s1 = _mk("admit_cfg", tables, maxcat=[1], gpucacheper=0.5, defaults=[0.0], max_batch=8192,
                          extra={"gpucache_admission": False})
    r1 = keys[cache1.query(0, keys) >= 0]
    c1k = keys[cache1.query(0, keys) < 0]
    s1.lookup(r1.astype(np.int64), [r1.size])
    q = c1k[:400].astype(np.int64)
    s1.lookup(q, [q.size])
    assert (cache1.query(0, q) >= 0).mean() > 0.97


def test_keys_narrow_as_offsets_from_each_tables_smallest_key():
    """Frame of reference (csrc/cache/key_pack.h): a request whose keys start high — ids with a per-table offset, here 2^40 + ...,
    2^33 + ... and a table of negative ids — crosses PCIe at 3 or 4 bytes per key like ids that start at 0, misses are fetched
    by their full keys, a key below a table's smallest key sends the call down the 8-byte path, and every row is exact."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(4040)
    R = 60000
    bases = [1 << 40, (1 << 33) + 12345, -(1 << 35)]
    tables = []
    for t, (b, D) in enumerate(zip(bases, [32, 128, 8])):
        k = (b + rng.permutation(3 * R)[:R]).astype(np.int64)
        tables.append((k, O.np_synth_rows(7, t, k, D)))
    ps, cache, s = _mk("forkeys", tables, maxcat=[1, 1, 1], gpucacheper=0.3, defaults=[0.5, 1.5, 2.5], max_batch=65536)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    nk = [60000, 50000, 40000]                      # > 128 K keys in all: the staged (narrowing) path
    for it in range(3):
        q = np.concatenate([rng.choice(k, n) for (k, _), n in zip(tables, nk)]).astype(np.int64)
        out = s.lookup(q, nk).cpu().numpy()
        st = s.last_stats()
        assert np.array_equal(_bits(out), _bits(co.lookup(q, nk, [0.5, 1.5, 2.5]))), it
        assert st.keys_narrowed == 1 and st.key_bytes == 3 and st.misses > 0, (it, st.key_bytes)
    # absent keys inside the frame (default rows) keep the width; one key below table 1's base widens the call
    q = np.concatenate([rng.choice(k, n) for (k, _), n in zip(tables, nk)]).astype(np.int64)
    q[5] = tables[0][0].min() + 3 * R + 17           # in nobody's table, inside table 0's frame
    out = s.lookup(q, nk).cpu().numpy()
    assert np.array_equal(_bits(out), _bits(co.lookup(q, nk, [0.5, 1.5, 2.5]))) and s.last_stats().key_bytes == 3
    q[nk[0] + 7] = tables[1][0].min() - 1
    out = s.lookup(q, nk).cpu().numpy()
    assert np.array_equal(_bits(out), _bits(co.lookup(q, nk, [0.5, 1.5, 2.5]))) and s.last_stats().key_bytes == 8


@pytest.mark.usefixtures("plain_lru")
@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("defer", [1, 0])
@pytest.mark.parametrize("base_misses", [40, 2500])   # 0.1-0.2 MB of missed rows: read in place; 4-5 MB: uploaded, scattered on the second stream
def test_insert_left_behind_the_call_is_visible_to_every_later_observer(direct, defer, base_misses):
    """Option defer_insert (default 1): a synchronous call returns when its rows are complete and leaves its cache-insert
    kernel enqueued behind it.  Everything that can observe the cache afterwards — the next lookup (this session's or another
    one's), hps_cache_query, the counters — sees the insert done; rows and counts equal those of defer_insert=0.
    Big requests (> 128 K keys) with few misses take the side-stream scatter next to the hit gather (host-gather tier)."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    rng = np.random.default_rng(40 + defer + 2 * int(direct) + base_misses)
    T, R, D = 3, 60000, 128
    tables = make_tables([(R, D)] * T)
    # (buckets filled to a quarter: no bucket is full, so that "every missed key is resident afterwards" holds to the key)
    ps, cache, s0 = _mk(f"defer{defer}{int(direct)}_{base_misses}", tables, maxcat=[1] * T, gpucacheper=0.5, max_batch=60000,
                        extra={"ps_direct_access": direct, "gpucache_load_factor": 0.25,
                               "gpucache_small_miss_insert_interval": 1})   # (every call inserts: this test watches the insert)
    s1 = hps.LookupSession.create(ps, f"defer{defer}{int(direct)}_{base_misses}", cache)
    for s in (s0, s1):
        s.set_option("defer_insert", defer)
        s.set_option("timing", 1)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    inserted0 = cache.counters()["inserted"]
    total_new = 0
    for it in range(6):
        sess = (s0, s1)[it % 2]
        nk = [50000, 47000, 50001]                       # 147,001 keys: a "big" request (1,024-key tiles, side-stream scatter)
        parts, missing = [], []
        for t, ((tk, _), n) in enumerate(zip(tables, nk)):
            res = tk[cache.query(t, tk) >= 0]
            cold = tk[cache.query(t, tk) < 0]
            q = rng.choice(res, n)
            m = rng.choice(cold, base_misses + 10 * it, replace=False)   # missed rows, each sent three times
            pos = rng.choice(n, 3 * m.size, replace=False)
            q[pos] = np.tile(m, 3)
            parts.append(q)
            missing.append(m)
        q = np.concatenate(parts).astype(np.int64)
        out = sess.lookup(q, nk).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(co.lookup(q, nk, [0.0] * T))), it
        st = sess.last_stats()
        new = sum(m.size for m in missing)
        assert (st.misses, st.unique_misses) == (3 * new, new), it
        total_new += new
        # observers: the query (device-synchronous), the counters, and the OTHER session's next lookup
        for t, m in enumerate(missing):
            assert (cache.query(t, m) >= 0).all(), (it, t)
        assert cache.counters()["inserted"] - inserted0 == total_new, it
        other = (s0, s1)[(it + 1) % 2]
        q2 = np.concatenate(missing).astype(np.int64)
        out2 = other.lookup(q2, [m.size for m in missing]).cpu().numpy()
        assert other.last_stats().misses == 0, it
        assert np.array_equal(_bits(out2), _bits(co.lookup(q2, [m.size for m in missing], [0.0] * T))), it
    assert s0.last_stats().insert_ms >= 0.0


@pytest.mark.parametrize("interval,admission", [(4, True), (2, True), (1, True), (4, False)])
def test_small_miss_calls_insert_every_nth_time(interval, admission):
    """ps.json gpucache_small_miss_insert_interval (default 4): a big request that missed only a few rows (they are read where the
    host gathered them) serves them exactly and inserts them only every n-th such call of its session — near-all-hit traffic does
    not pay a writer window per call.  What is left out is counted as dropped; with the admission rule off every call inserts."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    rng = np.random.default_rng(interval)
    T, R, D = 2, 90000, 64
    tables = make_tables([(R, D)] * T, seed=77)
    name = f"smi{interval}{int(admission)}"
    ps, cache, s0 = _mk(name, tables, maxcat=[1] * T, gpucacheper=0.5, max_batch=80000,
                        extra={"gpucache_load_factor": 0.25, "gpucache_small_miss_insert_interval": interval, "gpucache_admission": admission})
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    c0 = cache.counters()
    every = interval if admission else 1
    expect_inserted = expect_dropped = 0
    for it in range(1, 9):
        nk = [75000, 70000]                                # a "big" request (> 128 K keys)
        parts, missing = [], []
        for t, ((tk, _), n) in enumerate(zip(tables, nk)):
            res = tk[cache.query(t, tk) >= 0]
            cold = tk[cache.query(t, tk) < 0]
            q = rng.choice(res, n)
            m = rng.choice(cold, 30 + it, replace=False)
            q[rng.choice(n, m.size, replace=False)] = m
            parts.append(q)
            missing.append(m)
        q = np.concatenate(parts).astype(np.int64)
        out = s0.lookup(q, nk).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(co.lookup(q, nk, [0.0] * T))), it     # exact rows whether inserted or not
        new = sum(m.size for m in missing)
        inserts = it % every == 0
        expect_inserted += new if inserts else 0
        expect_dropped += 0 if inserts else new
        for t, m in enumerate(missing):
            assert bool((cache.query(t, m) >= 0).all()) == inserts and bool((cache.query(t, m) >= 0).any()) == inserts, (it, t)
        c = cache.counters()
        assert c["inserted"] - c0["inserted"] == expect_inserted and c["dropped"] - c0["dropped"] == expect_dropped, (it, c)
        assert c["unique_misses"] - c0["unique_misses"] == expect_inserted + expect_dropped
    s0.close()


def test_calls_that_miss_much_insert_every_time_whatever_the_interval():
    """Advisor finding of round 5: the small-miss interval looked only at the BYTES a call missed.  Narrow rows (16 B) make 14,000
    missed rows of a 140,000-key request 'few' (224 KB, read in place) — a call at 90 % hit inserted every 4th time and warmed the
    cache four times slower than the reference, which inserts every missing key below its hit_rate_threshold
    (docs/architecture.md:65-67).  Only near-all-hit calls (at most one key in 64 missed) skip now."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(5)
    R, D = 400000, 4
    tables = make_tables([(R, D)], seed=78)
    ps, cache, s0 = _mk("coldins", tables, maxcat=[1], gpucacheper=0.3, max_batch=150000, extra={"gpucache_load_factor": 0.25})
    co = O.COracle()
    co.add_table_arrays(*tables[0])
    tk = tables[0][0]
    c0 = cache.counters()
    for it in range(1, 5):
        res = tk[cache.query(0, tk) >= 0]
        cold = tk[cache.query(0, tk) < 0]
        n = 140000
        q = rng.choice(res, n)
        m = rng.choice(cold, 14000, replace=False)
        q[rng.choice(n, m.size, replace=False)] = m
        out = s0.lookup(q.astype(np.int64), [n]).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(co.lookup(q.astype(np.int64), [n], [0.0]))), it
        # (a key whose bucket is full of recently hit keys stays out under the admission rule: a handful, not three calls in four)
        assert (cache.query(0, m) >= 0).mean() > 0.98, f"call {it}: missed keys of a call at 90 % hit were left uncached"
        c = cache.counters()
        assert c["inserted"] - c0["inserted"] > 0.98 * 14000 * it and c["dropped"] - c0["dropped"] < 0.02 * 14000 * it, (it, c)
    s0.close()


def test_staged_keys_pulled_by_a_kernel_at_every_width_and_alignment():
    """Session option "keys_by_kernel": the staged keys of a big request are read out of the page-locked staging buffer by a kernel
    (hps_pull_bytes) instead of copy-engine copies.  Ragged tables (group offsets that are not multiples of 16 bytes at 3 bytes per
    key), 3-byte / uint32 / 8-byte keys: same rows as the oracle and as the copies."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(77)
    sizes = [(90001, 8), (70003, 4), (50000, 16)]
    k0 = rng.permutation(1 << 20)[:sizes[0][0]].astype(np.int64)
    k1 = (1 << 33) + rng.permutation(1 << 22)[:sizes[1][0]].astype(np.int64)           # high base, small offsets
    k2 = rng.permutation(1 << 28)[:sizes[2][0]].astype(np.int64) * 9                       # offsets beyond 24 bits
    tables = [(k, rng.standard_normal((k.size, d), dtype=np.float32)) for k, (_, d) in zip((k0, k1, k2), sizes)]
    ps, cache, s = _mk("pullkeys", tables, maxcat=[1, 1, 1], gpucacheper=0.5, max_batch=100000)
    try:
        for by_kernel in (1, 0, 2):
            s.set_option("keys_by_kernel", by_kernel)
            for nk, wide in (([70001, 65537, 0], 3), ([33333, 70003, 40001], 4), ([99999, 1, 50000], 8)):
                q = np.concatenate([rng.choice(tables[t][0], nk[t]) for t in range(3)]).astype(np.int64)
                if wide == 8:
                    q[12345] = -3
                    s.set_option("narrow_keys", 0)
                out = s.lookup(q, nk).cpu().numpy()
                s.set_option("narrow_keys", 1)
                assert s.last_stats().key_bytes == wide, (nk, wide, s.last_stats().key_bytes)
                assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, nk, [0.0] * 3))), (by_kernel, nk, wide)
    finally:
        s.close()
        ps.close()


@pytest.mark.parametrize("probe_in_lane", [2, 1, 0], ids=["probe_next_to_the_other_gather_while_missing_little", "probe_in_the_lane", "probe_never_in_the_lane"])
def test_two_sessions_near_all_hit_stress_rows_stay_exact(probe_in_lane):
    """Two sessions on one cache, big requests that miss a few hundred rows each (the regime of a production cache at
    99.9 % hit): side-stream scatter next to the hit gather, inserts left behind the calls, the other session's probes
    ordered behind them by the writer event — every row of every call exact, nothing lost from the counters.
    Session option "probe_in_lane": where K_P runs relative to the other session's K_G is scheduling only (2, the default:
    next to it while the session's calls miss little; 1: always behind it; 0: never)."""
    import threading
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    T, R, D = 2, 200000, 128
    tables = make_tables([(R, D)] * T)
    ps, cache, s0 = _mk("nearhit", tables, maxcat=[1] * T, gpucacheper=0.25, max_batch=100000)
    s1 = hps.LookupSession.create(ps, "nearhit", cache)
    for s_ in (s0, s1):
        s_.set_option("probe_in_lane", probe_in_lane)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    res = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
    cold = [tk[cache.query(t, tk) < 0] for t, (tk, _) in enumerate(tables)]
    errs = []
    sent_new = [0, 0]

    def worker(si, sess):
        rng = np.random.default_rng(900 + si)
        try:
            for it in range(12):
                nk = [100000, 90000]
                parts = []
                for t in range(T):
                    q = rng.choice(res[t], nk[t])
                    m = rng.choice(cold[t], 150)
                    q[rng.choice(nk[t], m.size, replace=False)] = m
                    parts.append(q)
                q = np.concatenate(parts).astype(np.int64)
                out = sess.lookup(q, nk).cpu().numpy()
                if not np.array_equal(_bits(out), _bits(co.lookup(q, nk, [0.0] * T))):
                    errs.append(f"session {si} call {it}: rows differ")
                sent_new[si] += sess.last_stats().unique_misses
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    c0 = cache.counters()
    th = [threading.Thread(target=worker, args=(i, s)) for i, s in enumerate((s0, s1))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    c1 = cache.counters()
    # every unique miss of every call went through an insert launch and was counted (inserted, refreshed — the other
    # session got there first — or dropped by the admission rule)
    done = sum(c1[k] - c0[k] for k in ("inserted", "refreshed", "dropped"))
    assert done == sum(sent_new), (done, sent_new)


def test_never_used_slots_of_a_cold_cache_stay_claimable_whatever_the_clock_reads(monkeypatch):
    """Advisor finding of round 3: never-used slots carried the stamp 128, a legal clock value — while the clock (or the
    stamp of newly inserted keys) read 128, they looked "written in the current unit" and no miss could take one: a cold
    cache (init_ec=false) dropped every miss that landed in a fresh bucket for a whole unit.  Free slots now carry a value
    the clock never takes.  Call clock with one call per unit: the clock reads 128 at call 128, the insert stamp at call 160."""
    monkeypatch.setenv("HPS_LRU_AGE_SHIFT", "0")
    monkeypatch.setenv("HPS_LRU_ADMIT", "0")
    from oracle import hps_oracle as O
    tables = make_tables([(60000, 16)])
    ps, cache, s = _mk("coldfree", tables, maxcat=[1], gpucacheper=1.0, max_batch=4096, extra={"init_ec": False})
    keys = tables[0][0]
    assert (cache.query(0, keys[:1000]) < 0).all()            # cold
    for call in range(200):
        q = keys[call * 50:(call + 1) * 50]                   # 50 keys nobody has asked for yet
        out = s.lookup(q, [50]).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, [50], [0.0]))), call
        assert s.last_stats().misses == 50, call
        slots = cache.query(0, q)
        assert (slots >= 0).all(), f"call {call}: {int((slots < 0).sum())} of 50 missed keys found no slot in a cache that is {call * 50 / 80000:.0%} full"
    c = cache.counters()
    assert c["inserted"] == 200 * 50 and c["dropped"] == 0


def test_missed_rows_go_up_in_growing_pieces_across_tables_of_any_width():
    """The host-gather tier uploads a call's missed rows in pieces of 0.5, 1, 2, then 4 MB (engine.cpp HandleMisses): a piece ends where
    its size is reached — in the middle of a table or across several — whatever the row width.  Tables of D = 1 / 3 / 100 / 128 / 1024 with
    nearly every key of a big call missing (a cold, small cache), twice (the second call finds what the first inserted), rows bit-exact."""
    from oracle import hps_oracle as O
    rng = np.random.default_rng(5120)
    spec = [(60000, 1), (50000, 3), (40000, 100), (70000, 128), (9000, 1024)]
    tables = make_tables(spec)
    maxcat = [6, 5, 4, 7, 1]
    ps, cache, s = _mk("ramp", tables, maxcat=maxcat, defaults=[0.5, -0.5, 1.5, 0.0, 2.0], gpucacheper=0.05, max_batch=8192)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    for it in range(2):
        nk = [8192 * m for m in maxcat]
        q = _queries(rng, tables, nk, miss_frac=0.02)
        out = s.lookup(q, nk).cpu().numpy()
        ref = co.lookup(q, nk, [0.5, -0.5, 1.5, 0.0, 2.0])
        assert np.array_equal(_bits(out), _bits(ref)), it
        st = s.last_stats()
        # (a cold 5-% cache: most of the call's distinct keys are missed rows — tens of MB in the 1024-wide and 128-wide tables)
        assert st.unique_misses > 60000, st.unique_misses
