"""Randomised parity sweep: random model shapes, cache sizes, request shapes and key mixes, both parameter-server
tiers, synchronous policy (exact rows) and per-table mixed policy — every answer bit-compared with the CPU oracle.
Seeds are fixed: a failure names the case and reproduces.
"""
import numpy as np
import pytest

from tests.conftest import make_tables
from tests.test_gpu_lookup import _bits, _mk

pytestmark = pytest.mark.gpu

DIMS = [1, 2, 3, 4, 8, 12, 16, 31, 32, 64, 100, 128, 200, 256]


def _case(seed):
    rng = np.random.default_rng(seed)
    T = int(rng.integers(1, 9))
    shapes = [(int(rng.integers(1, 6000)), int(rng.choice(DIMS))) for _ in range(T)]
    return rng, T, shapes


@pytest.mark.parametrize("seed", list(range(12)))
@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_random_models_sync_exact(seed, direct):
    from oracle import hps_oracle as O
    rng, T, shapes = _case(1000 + seed)
    tables = make_tables(shapes, seed=seed)
    defaults = [float(np.float32(rng.uniform(-2, 2))) for _ in range(T)]
    maxcat = [int(rng.integers(1, 5)) for _ in range(T)]
    B = int(rng.integers(1, 600))
    ps, cache, s = _mk(f"fz{seed}{'d' if direct else 'h'}", tables, maxcat=maxcat, defaults=defaults,
                       gpucacheper=float(rng.choice([0.01, 0.1, 0.5, 1.0])), max_batch=B, extra={"ps_direct_access": direct})
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    for it in range(4):
        nk = [int(rng.integers(0, B * maxcat[t] + 1)) for t in range(T)]
        if it == 3:
            nk = [B * maxcat[t] for t in range(T)]                      # the largest request the session accepts
        parts = []
        for (keys, _), n in zip(tables, nk):
            style = rng.integers(0, 4)
            if style == 0:
                qq = rng.choice(keys, n)                                  # uniform over the table
            elif style == 1:
                qq = rng.choice(keys[: max(1, keys.size // 20)], n)       # hot head: many duplicates
            elif style == 2:
                qq = np.full(n, keys[rng.integers(0, keys.size)])         # one key repeated
            else:
                qq = np.where(rng.random(n) < 0.5, rng.choice(keys, n), -1 - rng.integers(0, 1 << 50, n))  # half absent
            parts.append(qq.astype(np.int64))
        q = np.concatenate(parts) if parts else np.zeros(0, np.int64)
        out = s.lookup(q, nk).cpu().numpy()
        ref = co.lookup(q, nk, defaults)
        assert out.shape == ref.shape
        assert np.array_equal(_bits(out), _bits(ref)), (seed, direct, it, shapes, nk)
        st = s.last_stats()
        assert st.unique_misses <= st.misses <= q.size


@pytest.mark.parametrize("seed", list(range(6)))
@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_random_models_mixed_policy(seed, direct):
    """hit_rate_threshold inside (0,1): per call, the oracle's per-table rule says which tables answer from the cache
    only (defaults for misses) and which exactly; residency is read from the cache right before the call."""
    from oracle import hps_oracle as O
    rng, T, shapes = _case(2000 + seed)
    shapes = [(max(r, 200), d) for r, d in shapes]
    tables = make_tables(shapes, seed=100 + seed)
    defaults = [float(np.float32(rng.uniform(-2, 2))) for _ in range(T)]
    thr = float(rng.choice([0.3, 0.6, 0.9]))
    ps, cache, s = _mk(f"fm{seed}{'d' if direct else 'h'}", tables, maxcat=[2] * T, defaults=defaults, gpucacheper=0.4,
                       hit_rate_threshold=thr, max_batch=512, extra={"ps_direct_access": direct})
    for it in range(4):
        cache.wait_async()                                               # the oracle needs a stable residency snapshot
        resident = [tk[cache.query(t, tk) >= 0] for t, (tk, _) in enumerate(tables)]
        nk = [int(rng.integers(0, 1025)) for _ in range(T)]
        parts = []
        for t, ((keys, _), n) in enumerate(zip(tables, nk)):
            p_hot = rng.uniform(0, 1)
            pool_hot = resident[t] if resident[t].size else keys
            qq = np.where(rng.random(n) < p_hot, rng.choice(pool_hot, n), rng.choice(keys, n))
            parts.append(qq.astype(np.int64))
        q = np.concatenate(parts) if parts else np.zeros(0, np.int64)
        modes = O.np_insert_modes(q, nk, resident, thr)
        out = s.lookup(q, nk).cpu().numpy()
        ref = O.np_lookup(tables, q, nk, defaults, resident=[resident[t] if modes[t] else None for t in range(T)])
        assert np.array_equal(_bits(out), _bits(ref)), (seed, direct, it, thr, modes, nk)
        assert s.last_stats().async_insert == int(any(modes))


@pytest.mark.parametrize("seed", list(range(4)))
@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_big_requests_under_random_return_path_options(seed, direct):
    """Round 4 gave a synchronous call several return paths (missed rows read in place / uploaded and scattered on the second
    stream / scattered in the kernel lane behind a drained stream; insert left behind the call or waited for; miss counts read
    before or after the hit gather).  Two sessions on one cache, requests above 128 K keys (the big-request tiles), the options
    redrawn before every call, the miss volume swept from nothing to half the request: every row exact, every unique miss
    accounted for by the insert statistics."""
    import threading
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    rng0 = np.random.default_rng(7000 + seed)
    T = int(rng0.integers(2, 5))
    D = [int(rng0.choice([16, 64, 128, 128])) for _ in range(T)]
    R = 120000
    tables = make_tables([(R, d) for d in D], seed=50 + seed)
    per_table = 150000 // T + 2000
    name = f"opt{seed}{'d' if direct else 'h'}"
    ps, cache, s0 = _mk(name, tables, maxcat=[1] * T, gpucacheper=0.3, max_batch=per_table, extra={"ps_direct_access": direct})
    s1 = hps.LookupSession.create(ps, name, cache)
    co = O.COracle()
    for k, r in tables:
        co.add_table_arrays(k, r)
    errs, uniq_seen = [], [0, 0]

    def worker(si, sess):
        rng = np.random.default_rng(8000 + 10 * seed + si)
        try:
            for it in range(10):
                sess.set_option("defer_insert", int(rng.integers(0, 2)))
                sess.set_option("in_place_kb", int(rng.choice([0, 64, 1024, 8192])))
                sess.set_option("side_scatter_mb", int(rng.choice([0, 1, 16, 256])))
                sess.set_option("split_probe", int(rng.integers(0, 2)))
                nk = [int(rng.integers(per_table - 4000, per_table)) for _ in range(T)]
                miss_frac = float(rng.choice([0.0, 0.0005, 0.01, 0.1, 0.5]))
                parts = []
                for t, ((keys, _), n) in enumerate(zip(tables, nk)):
                    hot = keys[: int(0.25 * R)]                    # mostly resident (the warm set), the rest anywhere in the table
                    q = rng.choice(hot, n)
                    m = rng.random(n) < miss_frac
                    q[m] = rng.choice(keys, int(m.sum()))          # (keys that exist: a key in no tier is served the default and never counted)
                    parts.append(q.astype(np.int64))
                q = np.concatenate(parts)
                out = sess.lookup(q, nk).cpu().numpy()
                ref = co.lookup(q, nk, [0.0] * T, threads=2)
                if not np.array_equal(_bits(out), _bits(ref)):
                    errs.append(f"session {si} call {it}: rows differ (miss_frac {miss_frac})")
                uniq_seen[si] += sess.last_stats().unique_misses
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    c0 = cache.counters()
    th = [threading.Thread(target=worker, args=(i, s)) for i, s in enumerate((s0, s1))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    c1 = cache.counters()
    assert sum(c1[k] - c0[k] for k in ("inserted", "refreshed", "dropped")) == sum(uniq_seen)
    s0.close()
    s1.close()
