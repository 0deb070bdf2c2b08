"""ps.json parsing of the engine vs the key list / defaults / tolerant conversions of the reference
(HPSBackend::ParseParameterServer, backend.cpp:102-526; TritonJsonHelper::parse, triton_helpers.cpp:42-442)."""
import copy
import json

import numpy as np
import pytest

from tests.conftest import ps_config


def _tables(T=2):
    return [(np.arange(4, dtype=np.int64), np.zeros((4, d), np.float32)) for d in ([1, 16] + [8] * T)[:T]]


def _mk(cfg):
    from hugectr_backend_amd import hps
    return hps.HierParameterServer.create_from_dict(cfg, load_tables=False)


def test_readme_sample_ps_json_parses_with_reference_defaults():
    """The sample of README.md:127-159 / Deployment.ipynb:280-316."""
    cfg = {
        "supportlonglong": True,
        "volatile_db": {"type": "hash_map", "user_name": "default", "num_partitions": 8, "max_get_batch_size": 100000,
                        "max_set_batch_size": 100000, "overflow_policy": "evict_oldest", "overflow_margin": 10000000,
                        "overflow_resolution_target": 0.8, "initial_cache_rate": 1.0},
        "persistent_db": {"type": "disabled"},
        "models": [{
            "model": "hps_wdl", "sparse_files": ["/a/wdl0_sparse_20000.model", "/a/wdl1_sparse_20000.model"],
            "num_of_worker_buffer_in_pool": 3, "embedding_table_names": ["embedding_table1", "embedding_table2"],
            "embedding_vecsize_per_table": [1, 16], "maxnum_catfeature_query_per_table_per_sample": [2, 26],
            "default_value_for_each_table": [0.0, 0.0], "deployed_device_list": [0], "max_batch_size": 1024,
            "cache_refresh_percentage_per_iteration": 0.2, "hit_rate_threshold": 0.9, "gpucacheper": 0.5, "gpucache": True}],
    }
    ps = _mk(cfg)
    mi = ps.get_hps_model_configuration_map()["hps_wdl"]
    assert (mi.max_batch_size, mi.num_tables, mi.use_gpu_embedding_cache) == (1024, 2, 1)
    assert mi.hit_rate_threshold == pytest.approx(0.9) and mi.cache_size_percentage == pytest.approx(0.5)
    assert (mi.number_of_worker_buffers_in_pool, mi.number_of_refresh_buffers_in_pool) == (3, 1)
    assert mi.cache_refresh_percentage_per_iteration == pytest.approx(0.2)
    assert (mi.cat_num, mi.embedding_size, mi.device_id) == (28, 17, 0)   # model_state.cpp:337-356, backend.cpp:422
    t1 = ps.table_info("hps_wdl", 1)
    assert (t1.embedding_vecsize, t1.maxnum_catfeature, t1.default_value) == (16, 26, 0.0)


def test_string_encoded_numbers_and_bools_are_accepted():
    """triton_helpers.cpp:47-60,75-79,133-137."""
    cfg = ps_config("m", _tables(), gpucache=True)
    m = cfg["models"][0]
    cfg["supportlonglong"] = "True"
    m.update(max_batch_size="512", gpucache="1", hit_rate_threshold="0.75", gpucacheper="0.25",
             num_of_worker_buffer_in_pool="4", deployed_device_list=["0", 1], default_value_for_each_table=["0.5", 2],
             embedding_vecsize_per_table=["1", 16], maxnum_catfeature_query_per_table_per_sample=[2, "26"])
    mi = _mk(cfg).model_info("m")
    assert (mi.max_batch_size, mi.use_gpu_embedding_cache, mi.number_of_worker_buffers_in_pool) == (512, 1, 4)
    assert mi.hit_rate_threshold == pytest.approx(0.75) and mi.device_id == 1 and mi.num_deployed_devices == 2


@pytest.mark.parametrize("key", ["model", "max_batch_size", "sparse_files", "gpucache", "num_of_worker_buffer_in_pool",
                                 "deployed_device_list", "default_value_for_each_table",
                                 "maxnum_catfeature_query_per_table_per_sample", "embedding_vecsize_per_table",
                                 "hit_rate_threshold", "gpucacheper"])
def test_mandatory_model_keys(key):
    """required flags of backend.cpp:318-460 (hit_rate_threshold / gpucacheper only when gpucache is on)."""
    from hugectr_backend_amd import hps
    cfg = ps_config("m", _tables(), gpucache=True)
    del cfg["models"][0][key]
    with pytest.raises(hps.HpsError) as e:
        _mk(cfg)
    assert e.value.code == hps.ERR_INVALID_ARG and f"'{key}' is mandatory" in e.value.msg


def test_cache_keys_are_optional_without_gpu_cache_and_supportlonglong_is_mandatory():
    from hugectr_backend_amd import hps
    cfg = ps_config("m", _tables(), gpucache=False)
    assert "hit_rate_threshold" not in cfg["models"][0]
    _mk(cfg)
    del cfg["supportlonglong"]
    with pytest.raises(hps.HpsError) as e:
        _mk(cfg)
    assert "supportlonglong" in e.value.msg
    cfg["supportlonglong"] = False   # only 64-bit keys are supported, as in the reference (model_state.cpp:213-218)
    with pytest.raises(hps.HpsError) as e:
        _mk(cfg)
    assert e.value.code == hps.ERR_UNSUPPORTED


@pytest.mark.parametrize("alias", ["hash_map", "hashmap", "hash", "map", "parallel_hash_map", "parallel-hashmap",
                                   "Parallel Hash", "redis_cluster", "redis", "rocks_db", "rocksdb", "disabled", "none"])
def test_database_type_aliases(alias):
    from hugectr_backend_amd import hps
    cfg = ps_config("m", _tables(), gpucache=False)
    cfg["volatile_db"]["type"] = alias   # triton_helpers.cpp:190-242 (space/dash -> underscore, lower case)
    canonical = {"redis": "redis_cluster", "rocks": "rocks_db", "disab": "disabled", "none": "disabled"}
    want = next((v for k, v in canonical.items() if alias.lower().startswith(k)), None)
    if want is None:
        _mk(cfg)  # the in-process host tier
    else:
        # the alias parses to the right enum, and a tier this build does not have is refused, not ignored
        with pytest.raises(hps.HpsError) as e:
            _mk(cfg)
        assert e.value.code == hps.ERR_UNSUPPORTED and want in e.value.msg


def test_unimplemented_tiers_are_refused_loudly():
    from hugectr_backend_amd import hps
    base = ps_config("m", _tables(), gpucache=False)
    for patch, needle in [({"persistent_db": {"type": "hash_map", "path": "/tmp/x"}}, "persistent_db"),
                          ({"update_source": {"type": "kafka", "brokers": "h:9092"}}, "update_source")]:
        cfg = copy.deepcopy(base)
        cfg.update(patch)
        with pytest.raises(hps.HpsError) as e:
            _mk(cfg)
        assert e.value.code == hps.ERR_UNSUPPORTED and needle in e.value.msg
    ok = copy.deepcopy(base)
    ok["persistent_db"] = {"type": "disabled"}
    _mk(ok)
    # the persistent tier and the bounded volatile tier exist (tests/test_host_tier_bounded.py): accepted
    ok["persistent_db"] = {"type": "rocksdb", "path": "/tmp/x", "read_only": True}
    ok["volatile_db"] = {"type": "hash_map", "initial_cache_rate": 0.5, "overflow_margin": 1000}
    _mk(ok)


def test_bad_enums_and_list_lengths_are_rejected():
    from hugectr_backend_amd import hps
    cfg = ps_config("m", _tables(), gpucache=False)
    bad = copy.deepcopy(cfg)
    bad["volatile_db"]["type"] = "memcached"
    with pytest.raises(hps.HpsError) as e:
        _mk(bad)
    assert "DatabaseType_t" in e.value.msg
    bad = copy.deepcopy(cfg)
    bad["volatile_db"]["overflow_policy"] = "evict_newest"
    with pytest.raises(hps.HpsError) as e:
        _mk(bad)
    assert "DatabaseOverflowPolicy_t" in e.value.msg
    for key in ("embedding_vecsize_per_table", "default_value_for_each_table", "maxnum_catfeature_query_per_table_per_sample"):
        bad = copy.deepcopy(cfg)
        bad["models"][0][key] = bad["models"][0][key][:1]
        with pytest.raises(hps.HpsError) as e:
            _mk(bad)
        assert key in e.value.msg
    bad = copy.deepcopy(cfg)
    bad["models"][0]["max_batch_size"] = "many"
    with pytest.raises(hps.HpsError):
        _mk(bad)


def test_default_table_names_and_json_syntax_errors(tmp_path):
    from hugectr_backend_amd import hps
    with pytest.raises(hps.HpsError) as e:
        hps.HierParameterServer.create(str(tmp_path / "nope.json"))
    assert "cannot open" in e.value.msg
    p = tmp_path / "broken.json"
    p.write_text('{"supportlonglong": true, "models": [')
    with pytest.raises(hps.HpsError) as e:
        hps.HierParameterServer.create(str(p))
    assert "JSON parse error" in e.value.msg
    # a model whose table directory does not exist fails the load with the path in the message
    cfg = ps_config("m", _tables(), gpucache=False)
    p.write_text(json.dumps(cfg))
    with pytest.raises(hps.HpsError) as e:
        hps.HierParameterServer.create(str(p))
    assert "/nonexistent/m_0" in e.value.msg


def test_engine_knobs_of_this_build_are_optional_and_typed():
    """ps.json keys the reference does not have: accepted when well-formed, refused when not."""
    from hugectr_backend_amd import hps
    cfg = ps_config("knobs", _tables(), gpucache=False)
    cfg["models"][0].update({"gpucache_admission": False, "gpucache_load_factor": 0.5})
    _mk(cfg)
    cfg["models"][0]["gpucache_admission"] = "true"          # the reference's tolerant conversions apply
    _mk(cfg)
    cfg["models"][0]["gpucache_admission"] = [1]
    with pytest.raises(hps.HpsError):
        _mk(cfg)
