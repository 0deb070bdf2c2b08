"""GPU cache on top of a host tier that is smaller than the table (bounded volatile database + persistent row store,
tests/test_host_tier_bounded.py): the three tiers together still return the oracle's rows bit for bit.
Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest

from tests.conftest import make_tables, ps_config

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _cfg(tmp_path, tables, vdb, **kw):
    from oracle import hps_oracle as O
    dirs = []
    for t, (k, r) in enumerate(tables):
        O.np_write_table(tmp_path / f"t{t}", k, r)
        dirs.append(str(tmp_path / f"t{t}"))
    cfg = ps_config("m", tables, dirs=dirs, **kw)
    cfg["volatile_db"].update(vdb)
    cfg["persistent_db"] = {"type": "rocks_db", "path": str(tmp_path / "store")}
    return cfg


@pytest.mark.parametrize("policy", ["evict_random", "evict_oldest", "evict_least_used"])
def test_three_tiers_return_exact_rows(tmp_path, policy):
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    rng = np.random.default_rng(len(policy))
    tables = make_tables([(20000, 16), (6000, 128)])
    vdb = {"overflow_margin": 150, "overflow_policy": policy, "overflow_resolution_target": 0.75,
           "initial_cache_rate": 0.05, "cache_missed_embeddings": True}
    cfg = _cfg(tmp_path, tables, vdb, gpucacheper=0.1, hit_rate_threshold=1.0, defaults=[0.5, -1.0], max_batch=4096)
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=True)     # loads, warms the GPU cache from the store
    cache = ps.get_embedding_cache("m", 0)
    s = hps.LookupSession.create(ps, "m", cache)
    for _ in range(12):
        nk = [int(rng.integers(1, 4096)), int(rng.integers(1, 4096))]
        parts = []
        for (k, _), n in zip(tables, nk):
            q = rng.choice(k, n)
            cold = rng.random(n) < 0.1
            parts.append(np.where(cold, -1 - rng.integers(0, 1 << 40, n), q).astype(np.int64))
        q = np.concatenate(parts)
        out = s.lookup(q, nk).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(O.np_lookup(tables, q, nk, [0.5, -1.0])))
    for t in range(2):
        st = ps.host_tier_stats("m", t)
        assert st["tiered"] == 1 and st["max_partition_entries"] <= 150
        assert st["persistent_hits"] > 0 and st["overflows"] > 0 and st["not_found"] > 0
    # refresh re-reads the resident keys through the bounded tier
    ps.upsert("m", 0, tables[0][0][:64], np.full((64, 16), 3.25, np.float32))
    ps.refresh_embedding_cache("m", 0)
    out = s.lookup(np.concatenate([tables[0][0][:64], tables[1][0][:1]]), [64, 1]).cpu().numpy()
    assert np.all(out[:64 * 16] == 3.25)


def test_device_driven_tier_needs_the_whole_table_in_ram(tmp_path):
    from hugectr_backend_amd import hps
    tables = make_tables([(2000, 8)])
    cfg = _cfg(tmp_path, tables, {"overflow_margin": 50}, gpucacheper=0.2, extra={"ps_direct_access": True})
    with pytest.raises(hps.HpsError, match="whole table in RAM"):
        hps.HierParameterServer.create_from_dict(cfg, load_tables=True)
