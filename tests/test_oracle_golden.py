"""The CPU oracle (C and NumPy restatements) against the committed fixtures in tests/golden/ and against each
other; the product's host tier against the same fixtures.  Runs without a GPU."""
from pathlib import Path

import numpy as np
import pytest

G = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def g():
    return np.load(G / "hps_golden.npz")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_generator_known_answers(g):
    from oracle import hps_oracle as O
    assert np.array_equal(O.np_mix64(np.arange(3, dtype=np.uint64)), g["mix64_of_0_1_2"])
    assert [int(O.COracle.lib().oracle_mix64(i)) for i in range(3)] == [int(x) for x in g["mix64_of_0_1_2"]]
    assert np.array_equal(_bits(O.np_synth_rows(O.SEED, 3, np.arange(5, 9), 16)), _bits(g["synth_t3_k5_d16"]))
    assert np.array_equal(_bits(O.c_synth_rows(O.SEED, 3, 5, 4, 16)), _bits(g["synth_t3_k5_d16"]))
    r = g["synth_t3_k5_d16"]
    assert np.isfinite(r).all() and (r >= 0.5).all() and (r < 1.0).all()


def test_wdl_shape_fixture_c_and_numpy(g):
    from oracle import hps_oracle as O
    tabs = [(g["wdl_k0"], g["wdl_r0"]), (g["wdl_k1"], g["wdl_r1"])]
    nk = g["wdl_numkeys"]
    assert g["wdl_expected"].shape == (4180,)  # 10*2*1 + 10*26*16 (Deployment.ipynb:793-795)
    assert np.array_equal(_bits(O.np_lookup(tabs, g["wdl_keys"], nk, g["wdl_defaults"])), _bits(g["wdl_expected"]))
    co = O.COracle()
    for k, r in tabs:
        co.add_table_arrays(k, r)
    for th in (1, 3):
        assert np.array_equal(_bits(co.lookup(g["wdl_keys"], nk, g["wdl_defaults"], threads=th)), _bits(g["wdl_expected"]))


def test_identity_table_files_written_with_struct_pack(g):
    """lookup(k) == row k for a table written by the notebook recipe (01_model_training.ipynb:498-504)."""
    from oracle import hps_oracle as O
    keys, rows = O.np_read_table(G / "identity_table", 4)
    assert np.array_equal(keys, np.arange(32)) and np.array_equal(_bits(rows), _bits(g["identity_rows"]))
    co = O.COracle()
    co.add_table_dir(G / "identity_table", 4)
    q = np.arange(32, dtype=np.int64)[::-1].copy()
    out = co.lookup(q, [32], [0.0]).reshape(32, 4)
    assert np.array_equal(_bits(out), _bits(g["identity_rows"][::-1]))
    with pytest.raises(OSError):
        O.COracle().add_table_dir(G / "identity_table", 5)  # emb_vector size must equal R*D*4


def test_default_fill_duplicates_order_and_file_duplicates(g):
    from oracle import hps_oracle as O
    ident = (np.arange(32, dtype=np.int64), g["identity_rows"])
    co = O.COracle()
    co.add_table_arrays(*ident)
    assert np.array_equal(_bits(co.lookup(g["default_keys"], [7], [1.0])), _bits(g["default_expected_1"]))
    assert np.array_equal(_bits(co.lookup(g["default_keys"], [7], [0.0])), _bits(g["default_expected_0"]))
    assert np.array_equal(_bits(co.lookup(g["dups_keys"], [9], [0.0])), _bits(g["dups_expected"]))
    e = g["dups_expected"].reshape(9, 4)
    assert np.array_equal(e[0], e[1]) and np.array_equal(e[0], e[3]) and np.array_equal(e[6], e[8])
    co2 = O.COracle()
    co2.add_table_arrays(g["filedup_keys"], g["filedup_rows"])
    out = co2.lookup(g["filedup_query"], [4], [2.5])
    assert np.array_equal(_bits(out), _bits(g["filedup_expected"]))
    assert np.array_equal(out.reshape(4, 2)[0], g["filedup_rows"][2])   # key 9: last row wins
    assert (out.reshape(4, 2)[3] == 2.5).all()


def test_tf_ensemble_shape_fixture(g):
    from oracle import hps_oracle as O
    co = O.COracle()
    co.add_table_arrays(g["tf_k"], g["tf_r"])
    out = co.lookup(g["tf_keys"], [3072], [1.0], threads=2)
    assert out.shape == (3072 * 16,)  # feeds a dense model expecting [1024, 48] (03_...ipynb:264-265)
    assert np.array_equal(_bits(out), _bits(g["tf_expected"]))


def test_c_and_numpy_oracles_agree_on_random_requests():
    from oracle import hps_oracle as O
    rng = np.random.default_rng(9)
    for trial in range(10):
        T = int(rng.integers(1, 5))
        tabs, co = [], O.COracle()
        for t in range(T):
            R, D = int(rng.integers(1, 400)), int(rng.choice([1, 2, 7, 16, 128]))
            k = rng.integers(-1000, 1000, R).astype(np.int64)  # duplicates on purpose: last wins
            r = O.np_synth_rows(trial, t, np.arange(R), D)
            tabs.append((k, r))
            co.add_table_arrays(k, r)
        nk = [int(rng.integers(0, 300)) for _ in range(T)]
        q = rng.integers(-1200, 1200, sum(nk)).astype(np.int64)
        df = rng.random(T).astype(np.float32)
        assert np.array_equal(_bits(co.lookup(q, nk, df, threads=int(rng.integers(1, 4)))), _bits(O.np_lookup(tabs, q, nk, df)))


def test_product_host_tier_against_fixtures(g, tmp_path):
    """gpucache=false lookup session of the product (CPU parameter server) on the golden vectors."""
    from hugectr_backend_amd import hps
    from tests.conftest import ps_config
    tabs = [(g["wdl_k0"], g["wdl_r0"]), (g["wdl_k1"], g["wdl_r1"])]
    ps = hps.HierParameterServer.create_from_dict(ps_config("wdl", tabs, gpucache=False, maxcat=[2, 26]), load_tables=False)
    for t, (k, r) in enumerate(tabs):
        ps.load_table_arrays("wdl", t, k, r)
    s = hps.LookupSession.create(ps, "wdl", None)
    out = s.lookup(g["wdl_keys"], g["wdl_numkeys"])
    assert np.array_equal(_bits(out), _bits(g["wdl_expected"]))
    # identity table from files, through ps.json
    cfg = ps_config("ident", [(np.arange(32), g["identity_rows"])], dirs=[str(G / "identity_table")], gpucache=False, defaults=[1.0])
    ps2 = hps.HierParameterServer.create_from_dict(cfg, load_tables=True)
    s2 = hps.LookupSession.create(ps2, "ident", None)
    assert np.array_equal(_bits(s2.lookup(g["default_keys"], [7])), _bits(g["default_expected_1"]))
    assert np.array_equal(_bits(s2.lookup(g["dups_keys"], [9])), _bits(g["dups_expected"]))


def test_insertion_policy_per_table_restatement():
    """np_insert_modes / np_lookup(resident=[.., None, ..]): the per-table sync/async rule the GPU tests check against."""
    from oracle import hps_oracle as O
    tk = np.arange(100, dtype=np.int64)
    rows = O.np_synth_rows(O.SEED, 0, tk, 4)
    tables = [(tk, rows), (tk, rows), (tk, rows)]
    resident = [tk[:90], tk[:50], tk]
    q = np.concatenate([tk, tk, tk[:10]])
    nk = [100, 100, 10]
    # table 0: 90 % hit, table 1: 50 %, table 2: no misses
    assert O.np_insert_modes(q, nk, resident, 0.8) == [True, False, False]
    assert O.np_insert_modes(q, nk, resident, 1.0) == [False, False, False]       # threshold 1.0: always synchronous
    assert O.np_insert_modes(q, nk, resident, 0.4) == [True, True, False]
    out = O.np_lookup(tables, q, nk, [9.0, 9.0, 9.0], resident=[resident[0], None, None]).reshape(-1, 4)
    assert (out[90:100] == 9.0).all() and np.array_equal(out[:90], rows[:90])     # async table: defaults for non-resident keys
    assert np.array_equal(out[100:200], rows)                                       # sync table: exact rows


def test_insertion_policy_counts_unique_keys_not_keys_as_sent():
    """SURVEY.md App. C4 / docs/hierarchical_parameter_server.md:69: the cache is queried with the batch's UNIQUE keys, so
    the hit rate behind the sync/async decision is 1 - unique misses / unique keys.  A skewed batch separates the two
    definitions: 900 copies of one resident key + 100 distinct cold keys is "90 % hit" over the keys as sent and
    1/101 = 1 % hit over unique keys."""
    from oracle import hps_oracle as O
    tk = np.arange(1000, dtype=np.int64)
    resident = [tk[:10]]
    q = np.concatenate([np.full(900, 3, dtype=np.int64), np.arange(500, 600, dtype=np.int64)])
    assert O.np_unique_counts(q, [1000], resident) == [(101, 100)]
    assert O.np_insert_modes(q, [1000], resident, 0.8) == [False]               # 1 % < 80 %: synchronous
    assert O.np_insert_modes_keys_as_sent(q, [1000], resident, 0.8) == [True]    # the round-1 reading says async
    # and the other way round: distinct resident keys + one cold key repeated
    q2 = np.concatenate([np.arange(10, dtype=np.int64), np.full(90, 700, dtype=np.int64)])
    assert O.np_insert_modes(q2, [100], resident, 0.8) == [True]                 # 10/11 = 91 % unique hit rate
    assert O.np_insert_modes_keys_as_sent(q2, [100], resident, 0.8) == [False]   # 10 % as sent
