"""GPU parity tests of the device-driven parameter-server tier ("ps_direct_access": true in ps.json).

Same bar as test_gpu_lookup.py (bit-exact against the CPU oracle); the difference under test is who resolves
the missed keys: a kernel probing a device-resident index and reading rows in place from pinned host memory,
instead of host threads gathering into a staging buffer.
"""
import numpy as np
import pytest

from tests.test_gpu_lookup import _bits, _mk, _queries
from tests.conftest import make_tables

pytestmark = pytest.mark.gpu

DIRECT = {"ps_direct_access": True}


@pytest.mark.parametrize("thr", [1.0, 0.999])  # 1.0: no mid-call count read-back; <1: decision read-back, then sync
def test_direct_mixed_dims_exact(thr):
    from oracle import hps_oracle as O
    rng = np.random.default_rng(41)
    shapes = [(3000, 1), (2000, 16), (5000, 128), (700, 3), (4000, 100), (1500, 256)]
    tables = make_tables(shapes)
    T = len(tables)
    ps, cache, s = _mk(f"dmix{int(thr * 1000)}", tables, maxcat=[2] * T, defaults=[0.5 * t for t in range(T)],
                       gpucacheper=0.3, max_batch=2048, extra=DIRECT, hit_rate_threshold=thr)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    for it in range(4):
        nk = [int(rng.integers(0, 4096)) for _ in range(T)]
        if it == 2:
            nk[1] = 0
        q = _queries(rng, tables, nk, miss_frac=0.1)
        out = s.lookup(q, nk).cpu().numpy()
        ref = co.lookup(q, nk, [0.5 * t for t in range(T)])
        assert np.array_equal(_bits(out), _bits(ref)), it
        st = s.last_stats()
        assert st.async_insert == 0
        assert st.unique_misses <= st.misses <= q.size
    assert cache.counters()["inserted"] > 0


def test_direct_miss_accounting_matches_host_path():
    """Same queries through a host-gather server and a direct server: same output bits, same miss counts."""
    rng = np.random.default_rng(42)
    tables = make_tables([(20000, 128)] * 4)
    _, ch, sh = _mk("acc_host", tables, maxcat=[1] * 4, gpucacheper=0.2, max_batch=4096)
    _, cd, sd = _mk("acc_direct", tables, maxcat=[1] * 4, gpucacheper=0.2, max_batch=4096, extra=DIRECT)
    nk = [4096] * 4
    q = _queries(rng, tables, nk, miss_frac=0.03)

    def expected(cache):  # (misses, unique misses) from the cache's own residency just before the call
        m = u = 0
        for t in range(4):
            qt = q[t * 4096:(t + 1) * 4096]
            absent = qt[cache.query(t, qt) < 0]
            m += absent.size
            u += np.unique(absent).size
        return m, u

    eh, ed = expected(ch), expected(cd)
    oh = sh.lookup(q, nk).cpu().numpy()
    od = sd.lookup(q, nk).cpu().numpy()
    assert np.array_equal(_bits(oh), _bits(od))
    a, b = sh.last_stats(), sd.last_stats()
    assert (a.misses, a.unique_misses) == eh
    assert (b.misses, b.unique_misses) == ed
    assert b.misses > 0


def test_direct_duplicate_table_keys_last_row_wins():
    """A key stored twice in the table file resolves to its last row, on the device index too."""
    from oracle import hps_oracle as O
    tables = make_tables([(2000, 16)])
    keys, rows = tables[0]
    keys = keys.copy()
    keys[1500:1600] = keys[100:200]
    tables = [(keys, rows)]
    ps, cache, s = _mk("ddup", tables, maxcat=[1], gpucacheper=0.05, max_batch=4096, extra=DIRECT)
    q = np.concatenate([keys[100:200], keys[:50], keys[1500:1600]]).astype(np.int64)
    out = s.lookup(q, [q.size]).cpu().numpy()
    ref = O.np_lookup(tables, q, [q.size], [0.0])
    assert np.array_equal(_bits(out), _bits(ref))
    assert np.array_equal(_bits(out.reshape(-1, 16)[:100]), _bits(rows[1500:1600]))


def test_direct_sentinel_key():
    from oracle import hps_oracle as O
    tables = make_tables([(500, 16)])
    keys, rows = tables[0]
    keys = keys.copy()
    keys[3] = np.iinfo(np.int64).min
    tables = [(keys, rows)]
    ps, cache, s = _mk("dsent", tables, maxcat=[1], gpucacheper=0.1, max_batch=1024, extra=DIRECT)
    q = np.array([keys[3], keys[4], np.iinfo(np.int64).min, np.iinfo(np.int64).max], dtype=np.int64)
    for _ in range(2):
        out = s.lookup(q, [4]).cpu().numpy()
        ref = O.np_lookup(tables, q, [4], [0.0])
        assert np.array_equal(_bits(out), _bits(ref))


def test_direct_eviction_pressure_and_device_keys():
    import torch
    from oracle import hps_oracle as O
    rng = np.random.default_rng(43)
    tables = make_tables([(30000, 64), (30000, 128)])
    ps, cache, s = _mk("devict", tables, maxcat=[1, 1], gpucacheper=0.01, max_batch=8192, extra=DIRECT)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    for it in range(5):
        nk = [8192, 8192]
        q = _queries(rng, tables, nk, miss_frac=0.02)
        if it % 2:
            out = s.lookup_device(torch.from_numpy(q).cuda(), nk).cpu().numpy()
        else:
            out = s.lookup(q, nk).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(co.lookup(q, nk, [0.0, 0.0]))), it


def test_direct_two_sessions_and_table_reload():
    """Two sessions hammer one direct cache while the host tier is reloaded under them (new model version);
    every answer must be exact against one of the two versions, and exact against the new one afterwards."""
    import threading
    from hugectr_backend_amd import hps
    tables = make_tables([(6000, 128), (6000, 16)])
    ps, cache, s0 = _mk("dshared", tables, maxcat=[1, 1], gpucacheper=0.05, max_batch=4096, extra=DIRECT)
    s1 = hps.LookupSession.create(ps, "dshared", cache)
    v1 = tables
    v2 = [(k, (r[::-1] * 2.0).astype(np.float32).copy()) for k, r in tables]
    from oracle import hps_oracle as O
    errs = []
    stop = threading.Event()

    def worker(sess, seed):
        rng = np.random.default_rng(seed)
        try:
            while not stop.is_set():
                nk = [2048, 2048]
                q = _queries(rng, v1, nk, miss_frac=0.05)
                out = sess.lookup(q, nk).cpu().numpy()
                r1 = O.np_lookup(v1, q, nk, [0.0, 0.0])
                r2 = O.np_lookup(v2, q, nk, [0.0, 0.0])
                # rows cached before the reload may be served until the refresh: per element either version
                ok = (_bits(out) == _bits(r1)) | (_bits(out) == _bits(r2))
                if not ok.all():
                    errs.append("mismatch")
                    return
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(s0, 100)), threading.Thread(target=worker, args=(s1, 200))]
    [t.start() for t in th]
    import time
    time.sleep(0.3)
    for t, (k, r) in enumerate(v2):
        ps.load_table_arrays("dshared", t, k, r)
    ps.refresh_embedding_cache("dshared", 0)
    time.sleep(0.3)
    stop.set()
    [t.join() for t in th]
    assert not errs, errs
    rng = np.random.default_rng(9)
    q = _queries(rng, v2, [4096, 4096], miss_frac=0.05)
    out = s0.lookup(q, [4096, 4096]).cpu().numpy()
    assert np.array_equal(_bits(out), _bits(O.np_lookup(v2, q, [4096, 4096], [0.0, 0.0])))


def test_direct_reload_drops_vanished_keys():
    from oracle import hps_oracle as O
    tables = make_tables([(1000, 16)])
    keys, rows = tables[0]
    ps, cache, s = _mk("dvanish", tables, maxcat=[1], gpucacheper=0.1, defaults=[9.0], max_batch=2048, extra=DIRECT)
    keep = np.ones(keys.size, bool)
    keep[100:300] = False
    ps.load_table_arrays("dvanish", 0, keys[keep], rows[keep])
    ps.refresh_embedding_cache("dvanish", 0)
    q = keys[:600].copy()
    out = s.lookup(q, [600]).cpu().numpy()
    ref = O.np_lookup([(keys[keep], rows[keep])], q, [600], [9.0])
    assert np.array_equal(_bits(out), _bits(ref))


@pytest.mark.usefixtures("plain_lru")   # watches the insert mechanics: every missed key is taken in (conftest.py)
def test_direct_async_mode_uses_background_inserter():
    """hit rate above the threshold: defaults now, insertion in the background (host inserter), as in the host path."""
    from oracle import hps_oracle as O
    tables = make_tables([(8000, 32)])
    keys, rows = tables[0]
    ps, cache, s = _mk("dasync", tables, maxcat=[1], gpucacheper=0.5, hit_rate_threshold=0.5, defaults=[7.0],
                       max_batch=4096, extra=DIRECT)
    resident0 = keys[cache.query(0, keys) >= 0]
    cold = keys[cache.query(0, keys) < 0][:300]
    q = np.concatenate([resident0[:3000], cold]).astype(np.int64)
    out = s.lookup(q, [q.size]).cpu().numpy()
    st = s.last_stats()
    assert st.async_insert == 1 and st.misses == cold.size
    assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, [q.size], [7.0], resident=[resident0])))
    cache.wait_async()
    assert (cache.query(0, cold) >= 0).mean() > 0.8   # (a bucket full of just-hit keys takes no insert: counted as dropped)


def test_host_gather_option_on_a_direct_cache():
    """Session option "host_gather": one session serves its misses the reference's way (host threads + H2D) while
    another uses the device-driven fetch, on the same ps_direct_access cache and pinned tables; both exact."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    rng = np.random.default_rng(61)
    tables = make_tables([(9000, 128), (5000, 20), (300, 1)])
    ps, cache, s_dev = _mk("dmixed_tiers", tables, maxcat=[1, 1, 1], gpucacheper=0.1, max_batch=4096, extra=DIRECT)
    s_host = hps.LookupSession.create(ps, "dmixed_tiers", cache)
    s_host.set_option("host_gather", 1)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    for it in range(4):
        nk = [4096, 3000, 200]
        q = _queries(rng, tables, nk, miss_frac=0.05)
        for sess in (s_host, s_dev):
            out = sess.lookup(q, nk).cpu().numpy()
            assert np.array_equal(_bits(out), _bits(co.lookup(q, nk, [0.0] * 3))), (it, sess is s_host)
