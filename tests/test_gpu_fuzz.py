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
