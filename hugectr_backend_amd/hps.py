"""ctypes binding of libhps_amd.so (include/hps_amd.h).

Host-side mirror of the three HugeCTR classes the reference backend drives
(/root/reference/docs/architecture.md:232-323): ``HierParameterServer.create`` →
``get_embedding_cache`` → ``LookupSession.create`` → ``lookup``.  Names, argument meaning and error
behaviour follow the reference; the arithmetic happens in the native library (HIP kernels for the GPU
cache, C++ for the host tier).  There is no Python fallback: if the library is missing, import fails.

torch is used only to hold device memory in the convenience wrappers; the C ABI takes raw pointers.
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Sequence

import numpy as np

_PKG = Path(__file__).resolve().parent
import os as _os  # noqa: E402
# more hardware queues than HIP's default of 4, so that concurrent lookup sessions do not share one;
# only effective if HIP has not been initialised in this process yet
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# HPS_AMD_LIB_DIR: an instrumented build of the same sources (sanitizer job), see hugectr_backend_amd/build.py
_LIBPATH = (Path(_os.environ["HPS_AMD_LIB_DIR"]) if _os.environ.get("HPS_AMD_LIB_DIR") else _PKG / "lib") / "libhps_amd.so"


class HpsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[hps error {code}] {msg}")
        self.code = code
        self.msg = msg


# TRITONSERVER_Error_Code + 1
ERR_UNKNOWN, ERR_INTERNAL, ERR_NOT_FOUND, ERR_INVALID_ARG, ERR_UNAVAILABLE, ERR_UNSUPPORTED, ERR_ALREADY_EXISTS = range(1, 8)


class ModelInfo(C.Structure):
    _fields_ = [
        ("max_batch_size", C.c_uint64), ("num_tables", C.c_uint32), ("use_gpu_embedding_cache", C.c_int32),
        ("hit_rate_threshold", C.c_float), ("cache_size_percentage", C.c_float), ("i64_input_key", C.c_int32),
        ("number_of_worker_buffers_in_pool", C.c_int32), ("number_of_refresh_buffers_in_pool", C.c_int32),
        ("cache_refresh_percentage_per_iteration", C.c_float), ("device_id", C.c_int32),
        ("num_deployed_devices", C.c_uint32), ("refresh_delay", C.c_float), ("refresh_interval", C.c_float),
        ("cat_num", C.c_uint64), ("embedding_size", C.c_uint64),
    ]


class TableInfo(C.Structure):
    _fields_ = [("embedding_vecsize", C.c_uint32), ("maxnum_catfeature", C.c_uint64), ("default_value", C.c_float),
                ("rows_loaded", C.c_uint64)]


class HostTierStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("tiered", "persistent_rows", "entries", "capacity", "max_partition_entries",
                                          "lookups", "hits", "persistent_hits", "not_found", "inserts", "evictions",
                                          "overflows")]


class CacheTableInfo(C.Structure):
    _fields_ = [("embedding_vecsize", C.c_uint32), ("num_buckets", C.c_uint64), ("capacity_rows", C.c_uint64)]


class CacheCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("lookups", "keys", "misses", "unique_misses", "inserted", "refreshed", "dropped", "async_calls")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class LookupStats(C.Structure):
    _fields_ = [("misses", C.c_uint64), ("unique_misses", C.c_uint64), ("async_insert", C.c_int32),
                ("probe_gather_ms", C.c_float), ("phase_ms", C.c_float * 4), ("gpu_call_ms", C.c_float),
                ("hit_gather_ms", C.c_float), ("unique_keys", C.c_uint64), ("key_stage_ms", C.c_float),
                ("scatter_ms", C.c_float), ("insert_ms", C.c_float), ("keys_narrowed", C.c_int32), ("key_bytes", C.c_int32),
                ("miss_much_mode", C.c_int32), ("interact_separate", C.c_int32), ("mode_flips", C.c_uint64)]


class RefreshStats(C.Structure):
    _fields_ = [("tables", C.c_uint64), ("tables_unchanged", C.c_uint64), ("tables_full", C.c_uint64), ("keys_dumped", C.c_uint64),
                ("keys_changed", C.c_uint64), ("rows_refreshed", C.c_uint64), ("row_bytes", C.c_uint64), ("seconds", C.c_double)]


class ShardEntryStats(C.Structure):
    _fields_ = [("keys", C.c_uint64), ("unique_keys", C.c_uint64), ("misses", C.c_uint64), ("unique_misses", C.c_uint64),
                ("bucket_ms", C.c_float), ("lookup_ms", C.c_float), ("expand_ms", C.c_float), ("key_stage_ms", C.c_float),
                ("num_shards", C.c_uint32), ("key_bytes", C.c_uint32), ("sent", C.c_uint64 * 64), ("passes", C.c_uint32 * 64),
                ("shard_ms", C.c_float * 64), ("dedup_level", C.c_uint32), ("transport", C.c_uint32), ("copied_bytes", C.c_uint64),
                ("copy_wait_ms", C.c_float * 64), ("dedup_flips", C.c_uint64)]


def _load() -> C.CDLL:
    if not _LIBPATH.exists():
        raise ImportError(
            f"{_LIBPATH} is missing: build it with `python -m hugectr_backend_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no Python/CPU fallback for the engine.")
    L = C.CDLL(str(_LIBPATH), mode=C.RTLD_GLOBAL)
    P, cp, i32, u32, u64, i64 = C.c_void_p, C.c_char_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_int64
    sig = {
        "hps_last_error": (cp, []),
        "hps_device_count": (C.c_int, []),
        "hps_server_create": (C.c_int, [cp, C.POINTER(P)]),
        "hps_server_create_from_text": (C.c_int, [cp, C.c_int, C.POINTER(P)]),
        "hps_server_destroy": (None, [P]),
        "hps_server_model_count": (C.c_int, [P]),
        "hps_server_model_name": (cp, [P, C.c_int]),
        "hps_server_model_info": (C.c_int, [P, cp, C.POINTER(ModelInfo)]),
        "hps_server_table_info": (C.c_int, [P, cp, u32, C.POINTER(TableInfo)]),
        "hps_server_deployed_device": (C.c_int, [P, cp, u32, C.POINTER(i32)]),
        "hps_server_parse_config": (C.c_int, [P, cp]),
        "hps_server_update_database_per_model": (C.c_int, [P, cp]),
        "hps_server_create_embedding_cache_per_model": (C.c_int, [P, cp]),
        "hps_server_destroy_embedding_cache_per_model": (C.c_int, [P, cp]),
        "hps_server_refresh_embedding_cache": (C.c_int, [P, cp, i32]),
        "hps_server_refresh_embedding_cache_ex": (C.c_int, [P, cp, i32, i32, C.POINTER(RefreshStats)]),
        "hps_server_get_embedding_cache": (C.c_int, [P, cp, i32, C.POINTER(P)]),
        "hps_server_load_table_arrays": (C.c_int, [P, cp, u32, P, P, u64, C.c_int]),
        "hps_server_load_table_synthetic": (C.c_int, [P, cp, u32, u64, i64, u64]),
        "hps_server_load_table_synthetic_shard": (C.c_int, [P, cp, u32, u64, i64, u64, u32, u32]),
        "hps_server_fetch": (C.c_int, [P, cp, u32, P, u64, P, P]),
        "hps_server_upsert": (C.c_int, [P, cp, u32, P, P, u64]),
        "hps_server_table_data": (C.c_int, [P, cp, u32, C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_float)),
                                            C.POINTER(u64)]),
        "hps_server_host_tier_stats": (C.c_int, [P, cp, u32, C.POINTER(HostTierStats)]),
        "hps_server_host_tier_keys": (C.c_int, [P, cp, u32, P, u64, C.POINTER(u64)]),
        "hps_cache_num_tables": (C.c_int, [P]),
        "hps_cache_on_device": (C.c_int, [P]),
        "hps_cache_refresh_rows_uploaded": (u64, [P]),
        "hps_wake_copy_engines": (C.c_int, [C.c_int, C.c_char_p, u64]),
        "hps_pool_numa_node": (C.c_int, []),
        "hps_pool_fast_overruns": (u64, []),
        "hps_bind_calling_thread": (C.c_int, []),
        "hps_session_create_from_cache": (C.c_int, [P, C.POINTER(P)]),
        "hps_cache_table_info": (C.c_int, [P, u32, C.POINTER(CacheTableInfo)]),
        "hps_cache_counters": (C.c_int, [P, C.POINTER(CacheCounters)]),
        "hps_cache_query": (C.c_int, [P, u32, P, u64, P]),
        "hps_cache_wait_async": (C.c_int, [P]),
        "hps_cache_release": (None, [P]),
        "hps_session_create": (C.c_int, [P, cp, P, C.POINTER(P)]),
        "hps_session_destroy": (None, [P]),
        "hps_session_lookup": (C.c_int, [P, P, P, P, C.c_size_t]),
        "hps_session_lookup_device": (C.c_int, [P, P, P, P, C.c_size_t]),
        "hps_session_last_stats": (C.c_int, [P, C.POINTER(LookupStats)]),
        "hps_session_set_option": (C.c_int, [P, cp, C.c_int]),
        "hps_update_message_encode": (C.c_int, [cp, u32, u32, P, P, u64, P, u64, C.POINTER(u64)]),
        "hps_server_update_source_stats": (C.c_int, [P, P]),
        "hps_server_update_source_drain": (C.c_int, [P, u32]),
        "hps_server_update_source_filtered": (C.c_int, [P, C.POINTER(u64)]),
        "hps_server_update_source_stop": (C.c_int, [P]),
        "hps_shard_owner": (u32, [i64, u32]),
        "hps_shard_bucket_workspace_bytes": (u64, [u64, u32]),
        "hps_shard_bucket_device": (C.c_int, [P, u64, u32, P, P, P, P, P]),
        "hps_shard_unpermute_device": (C.c_int, [P, P, u64, u32, P, P]),
        "hps_shard_unique_id": (C.c_int, [P]),
        "hps_shard_session_create": (C.c_int, [P, u32, u32, P, u64, C.POINTER(P)]),
        "hps_shard_group_create_local": (C.c_int, [u32, C.POINTER(P)]),
        "hps_shard_session_lookup_host": (C.c_int, [P, P, u64, P]),
        "hps_shard_session_last_timing": (C.c_int, [P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                    C.POINTER(u64), C.POINTER(C.c_int32)]),
        "hps_shard_group_destroy": (None, [P]),
        "hps_shard_session_create_local": (C.c_int, [P, P, u32, u64, C.POINTER(P)]),
        "hps_shard_session_lookup": (C.c_int, [P, P, u64, P]),
        "hps_shard_session_last_stats": (C.c_int, [P, C.POINTER(u64), C.POINTER(u32), P, u32]),
        "hps_shard_session_destroy": (None, [P]),
        "hps_server_get_shard_cache": (C.c_int, [P, cp, u32, C.POINTER(P)]),
        "hps_shard_entry_create": (C.c_int, [P, cp, i32, C.POINTER(P)]),
        "hps_shard_entry_destroy": (None, [P]),
        "hps_shard_entry_lookup": (C.c_int, [P, P, P, P, C.c_size_t]),
        "hps_shard_entry_lookup_device": (C.c_int, [P, P, P, P, C.c_size_t]),
        "hps_shard_entry_last_stats": (C.c_int, [P, C.POINTER(ShardEntryStats)]),
        "hps_shard_entry_set_option": (C.c_int, [P, cp, C.c_int]),
        "hps_shard_entry_shard_capacity": (u64, [P]),
        "hps_shard_plan_passes": (u64, [P, u32, u64, P, u64]),
        "hps_multi_gpu_selftest": (C.c_int, [P, u32, u64, u32, i32, C.c_char_p, u64]),
        "hps_dense_create": (C.c_int, [C.c_int, u32, u32, P, P, P, u32, u32, C.POINTER(P)]),
        "hps_dense_destroy": (None, [P]),
        "hps_dense_out_dim": (u32, [P]),
        "hps_dense_out_stride": (u32, [P]),
        "hps_dense_forward": (C.c_int, [P, P, P, u64, P, P]),
        "hps_session_lookup_interact_device": (C.c_int, [P, P, P, u64, P, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = symbol missing from the library
        fn.restype, fn.argtypes = res, args
    return L


LIB = _load()
EXPORTED_SYMBOLS = [
    "hps_last_error", "hps_device_count", "hps_server_create", "hps_server_create_from_text", "hps_server_destroy",
    "hps_server_model_count", "hps_server_model_name", "hps_server_model_info", "hps_server_table_info",
    "hps_server_deployed_device", "hps_server_parse_config", "hps_server_update_database_per_model",
    "hps_server_create_embedding_cache_per_model", "hps_server_destroy_embedding_cache_per_model",
    "hps_server_refresh_embedding_cache", "hps_server_refresh_embedding_cache_ex", "hps_server_get_embedding_cache", "hps_server_load_table_arrays",
    "hps_server_load_table_synthetic", "hps_server_load_table_synthetic_shard", "hps_server_fetch", "hps_server_upsert",
    "hps_server_table_data", "hps_update_message_encode", "hps_server_update_source_stats", "hps_server_update_source_drain", "hps_server_update_source_stop", "hps_server_update_source_filtered",
    "hps_cache_on_device", "hps_cache_refresh_rows_uploaded", "hps_wake_copy_engines", "hps_pool_numa_node", "hps_pool_fast_overruns", "hps_bind_calling_thread", "hps_session_create_from_cache",
    "hps_shard_unique_id", "hps_shard_session_create", "hps_shard_group_create_local", "hps_shard_group_destroy",
    "hps_shard_session_create_local", "hps_shard_session_lookup", "hps_shard_session_lookup_host", "hps_shard_session_last_timing",
    "hps_shard_session_last_stats", "hps_shard_session_destroy",
    "hps_server_host_tier_stats", "hps_server_host_tier_keys", "hps_cache_num_tables", "hps_cache_table_info",
    "hps_cache_counters", "hps_cache_query", "hps_cache_wait_async", "hps_cache_release", "hps_session_create",
    "hps_session_destroy", "hps_session_lookup", "hps_session_lookup_device", "hps_session_last_stats",
    "hps_session_set_option", "hps_shard_owner", "hps_shard_bucket_workspace_bytes", "hps_shard_bucket_device",
    "hps_shard_unpermute_device", "hps_dense_create", "hps_dense_destroy", "hps_dense_out_dim", "hps_dense_out_stride",
    "hps_dense_forward", "hps_session_lookup_interact_device",
    "hps_server_get_shard_cache", "hps_shard_entry_create", "hps_shard_entry_destroy", "hps_shard_entry_lookup",
    "hps_shard_entry_lookup_device", "hps_shard_entry_last_stats", "hps_shard_entry_set_option", "hps_shard_entry_shard_capacity",
    "hps_shard_plan_passes", "hps_multi_gpu_selftest",
]


def _check(rc: int):
    if rc != 0:
        raise HpsError(rc, (LIB.hps_last_error() or b"").decode(errors="replace"))


def device_count() -> int:
    return int(LIB.hps_device_count())


def pool_numa_node() -> int:
    """NUMA node the host tier's worker pools are bound to (-1: not bound); decided by the first server of the process."""
    return int(LIB.hps_pool_numa_node())


def pool_fast_overruns() -> int:
    """Fork-joins of the lock-free pool path whose tasks ran more than once (0 unless round 5's slot-reuse race is back)."""
    return int(LIB.hps_pool_fast_overruns())


def bind_calling_thread() -> bool:
    """The calling thread joins the worker pools' NUMA node (no-op when the pools are not bound or the thread is already placed)."""
    return bool(LIB.hps_bind_calling_thread())


def multi_gpu_selftest(devices, probe_bytes: int = 64 << 20, timeout_s: float = 20.0, with_rccl: bool = True) -> dict:
    """First contact with a multi-GPU machine (include/hps_amd.h: hps_multi_gpu_selftest): peer access matrix, a 4-KB peer store
    per ordered pair, GB/s per pair of kernel stores and of copy-engine copies, one RCCL all-reduce of one word — behind a
    deadline.  Always returns the report (dict); "timeout": True + "stuck_in" when a step did not come back."""
    import json
    devs = [int(d) for d in devices]
    arr = (C.c_int32 * len(devs))(*devs)
    buf = C.create_string_buffer(1 << 16)
    rc = int(LIB.hps_multi_gpu_selftest(arr, len(devs), int(probe_bytes), int(timeout_s * 1000), 1 if with_rccl else 0, buf, len(buf)))
    text = buf.value.decode(errors="replace")
    try:
        rep = json.loads(text) if text else {}
    except ValueError:
        rep = {"unparsed": text[:300]}
    if rc != 0 and not rep.get("timeout"):
        _check(rc)
    return rep


def wake_copy_engines(device: int = 0):
    """(engines that took a copy, one-line report): one tiny copy through every SDMA engine of the device in both
    directions, once per process (cache creation does it on its own; csrc/cache/copy_engines.h)."""
    buf = C.create_string_buffer(256)
    n = int(LIB.hps_wake_copy_engines(int(device), buf, 256))
    return n, buf.value.decode()


class EmbeddingCache:
    """HugeCTR::EmbeddingCacheBase handle (shared by every session of one model on one device)."""

    def __init__(self, handle, device: int = 0):
        self._h = handle
        self.device = int(device)
        self.on_device = True   # False: handle of a model that runs without GPU cache

    @property
    def num_tables(self) -> int:
        return int(LIB.hps_cache_num_tables(self._h))

    def table_info(self, t: int) -> CacheTableInfo:
        ti = CacheTableInfo()
        _check(LIB.hps_cache_table_info(self._h, t, C.byref(ti)))
        return ti

    def counters(self) -> dict:
        c = CacheCounters()
        _check(LIB.hps_cache_counters(self._h, C.byref(c)))
        return c.as_dict()

    def refresh_rows_uploaded(self) -> int:
        """Rows the refreshes of this cache have uploaded so far (live, piece by piece)."""
        return int(LIB.hps_cache_refresh_rows_uploaded(self._h))

    def query(self, table: int, keys) -> np.ndarray:
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty(keys.size, dtype=np.int32)
        _check(LIB.hps_cache_query(self._h, table, keys.ctypes.data, keys.size, out.ctypes.data))
        return out

    def wait_async(self):
        _check(LIB.hps_cache_wait_async(self._h))

    def release(self):
        if self._h:
            LIB.hps_cache_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class HierParameterServer:
    """HugeCTR::HierParameterServerBase (docs/architecture.md:246-274)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def create(cls, ps_json_config_file: str) -> "HierParameterServer":
        h = C.c_void_p()
        _check(LIB.hps_server_create(str(ps_json_config_file).encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def create_from_dict(cls, cfg: dict, load_tables: bool = True) -> "HierParameterServer":
        h = C.c_void_p()
        _check(LIB.hps_server_create_from_text(json.dumps(cfg).encode(), int(load_tables), C.byref(h)))
        return cls(h)

    def get_hps_model_configuration_map(self) -> dict:
        out = {}
        for i in range(LIB.hps_server_model_count(self._h)):
            name = LIB.hps_server_model_name(self._h, i).decode()
            out[name] = self.model_info(name)
        return out

    def model_info(self, model: str) -> ModelInfo:
        mi = ModelInfo()
        _check(LIB.hps_server_model_info(self._h, model.encode(), C.byref(mi)))
        return mi

    def table_info(self, model: str, t: int) -> TableInfo:
        ti = TableInfo()
        _check(LIB.hps_server_table_info(self._h, model.encode(), t, C.byref(ti)))
        return ti

    def parse_config(self, path: str):
        _check(LIB.hps_server_parse_config(self._h, str(path).encode()))

    def update_database_per_model(self, model: str):
        _check(LIB.hps_server_update_database_per_model(self._h, model.encode()))

    def create_embedding_cache_per_model(self, model: str):
        _check(LIB.hps_server_create_embedding_cache_per_model(self._h, model.encode()))

    def destory_embedding_cache_per_model(self, model: str):  # [sic] reference spelling
        _check(LIB.hps_server_destroy_embedding_cache_per_model(self._h, model.encode()))

    def refresh_embedding_cache(self, model: str, device: int, full: bool = False) -> dict:
        """Rows of resident keys are taken again from the host tier — by default only those that can differ (tables reloaded or
        keys updated since the cache last looked); full=True: every resident row, as the reference does.  Returns what it did."""
        st = RefreshStats()
        _check(LIB.hps_server_refresh_embedding_cache_ex(self._h, model.encode(), device, 1 if full else 0, C.byref(st)))
        return {k: getattr(st, k) for k, _ in RefreshStats._fields_}

    def get_embedding_cache(self, model: str, device: int):
        h = C.c_void_p()
        _check(LIB.hps_server_get_embedding_cache(self._h, model.encode(), device, C.byref(h)))
        if not h:
            return None
        c = EmbeddingCache(h, device)
        if not LIB.hps_cache_on_device(h):   # gpucache=false model: the handle carries server + model only
            c.on_device = False
        return c

    def get_shard_cache(self, model: str, shard: int):
        """The cache of shard `shard` of a table-sharded model (ps.json "table_sharding": "hash"), or None."""
        h = C.c_void_p()
        _check(LIB.hps_server_get_shard_cache(self._h, model.encode(), shard, C.byref(h)))
        return EmbeddingCache(h, -1) if h else None

    def load_table_arrays(self, model: str, table: int, keys, rows):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.ndim == 2 and rows.shape[0] == keys.size
        _check(LIB.hps_server_load_table_arrays(self._h, model.encode(), table, keys.ctypes.data, rows.ctypes.data,
                                                keys.size, 0))

    def load_table_synthetic(self, model: str, table: int, seed: int, key0: int, rows: int, shard: int = 0,
                             num_shards: int = 1):
        """keys key0..key0+rows-1 with the synthetic rows; num_shards > 1 keeps only the keys `shard` owns."""
        if num_shards > 1:
            _check(LIB.hps_server_load_table_synthetic_shard(self._h, model.encode(), table, seed, key0, rows, shard,
                                                             num_shards))
        else:
            _check(LIB.hps_server_load_table_synthetic(self._h, model.encode(), table, seed, key0, rows))

    def fetch(self, model: str, table: int, keys, return_found: bool = False):
        """Host-tier (volatile database) lookup of one table: rows or the table's default value."""
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        D = self.table_info(model, table).embedding_vecsize
        out = np.empty((keys.size, D), dtype=np.float32)
        found = np.empty(keys.size, dtype=np.uint8) if return_found else None
        _check(LIB.hps_server_fetch(self._h, model.encode(), table, keys.ctypes.data, keys.size, out.ctypes.data,
                                    found.ctypes.data if return_found else None))
        return (out, found) if return_found else out

    def table_data(self, model: str, table: int):
        """(keys[R], rows[R, D]) numpy views of the table as it sits in the host tier — no copy; valid until the table
        is reloaded or updated (tests / benchmarks only)."""
        kp, rp, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_float)(), C.c_uint64(0)
        _check(LIB.hps_server_table_data(self._h, model.encode(), table, C.byref(kp), C.byref(rp), C.byref(n)))
        R, D = int(n.value), int(self.table_info(model, table).embedding_vecsize)
        if R == 0:
            return np.zeros(0, np.int64), np.zeros((0, D), np.float32)
        return (np.ctypeslib.as_array(kp, shape=(R,)), np.ctypeslib.as_array(rp, shape=(R, D)))

    def upsert(self, model: str, table: int, keys, rows):
        """Online update of the host tier: insert-or-overwrite rows."""
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.size == keys.size * self.table_info(model, table).embedding_vecsize
        _check(LIB.hps_server_upsert(self._h, model.encode(), table, keys.ctypes.data, rows.ctypes.data, keys.size))

    def update_source_stats(self) -> dict:
        v = (C.c_uint64 * 6)()
        _check(LIB.hps_server_update_source_stats(self._h, v))
        return dict(zip(["messages", "keys", "dispatches", "commits", "dispatch_failures", "rejected_messages"], map(int, v)))

    def filtered_update_count(self) -> int:
        n = C.c_uint64(0)
        _check(LIB.hps_server_update_source_filtered(self._h, C.byref(n)))
        return int(n.value)

    def drain_update_source(self, timeout_ms: int = 10000):
        _check(LIB.hps_server_update_source_drain(self._h, int(timeout_ms)))

    def stop_update_source(self):
        _check(LIB.hps_server_update_source_stop(self._h))

    def host_tier_stats(self, model: str, table: int) -> dict:
        st = HostTierStats()
        _check(LIB.hps_server_host_tier_stats(self._h, model.encode(), table, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in HostTierStats._fields_}

    def host_tier_keys(self, model: str, table: int):
        """Keys the bounded volatile tier of the table holds now, ascending."""
        n = C.c_uint64(0)
        _check(LIB.hps_server_host_tier_keys(self._h, model.encode(), table, None, 0, C.byref(n)))
        out = np.empty(int(n.value), dtype=np.int64)
        _check(LIB.hps_server_host_tier_keys(self._h, model.encode(), table, out.ctypes.data, out.size, C.byref(n)))
        assert int(n.value) == out.size
        return out

    def close(self):
        if self._h:
            LIB.hps_server_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LookupSession:
    """HugeCTR::LookupSessionBase (docs/architecture.md:291-323)."""

    def __init__(self, handle, server: HierParameterServer, model: str):
        self._h = handle
        self._server = server  # keep alive
        self.model = model
        mi = server.model_info(model)
        self.num_tables = int(mi.num_tables)
        self.dims = [int(server.table_info(model, t).embedding_vecsize) for t in range(self.num_tables)]
        self.use_gpu_cache = bool(mi.use_gpu_embedding_cache)
        self.device = 0   # device of the session's embedding cache (create() sets it)

    @classmethod
    def create(cls, server: HierParameterServer, model: str, embedding_cache: EmbeddingCache | None) -> "LookupSession":
        h = C.c_void_p()
        _check(LIB.hps_session_create(server._h, model.encode(), embedding_cache._h if embedding_cache else None,
                                      C.byref(h)))
        s = cls(h, server, model)
        if embedding_cache is not None:
            s.device = embedding_cache.device
        return s

    @classmethod
    def create_from_cache(cls, server: HierParameterServer, model: str, embedding_cache: EmbeddingCache) -> "LookupSession":
        """LookupSessionBase::create(inference_params, embedding_cache) — the cache handle knows server and model."""
        h = C.c_void_p()
        _check(LIB.hps_session_create_from_cache(embedding_cache._h, C.byref(h)))
        s = cls(h, server, model)
        s.device = embedding_cache.device
        return s

    # -- the reference signature: lists of per-table pointers ---------------------------------------
    def lookup_ptrs(self, h_keys_ptrs: Sequence[int], vec_ptrs: Sequence[int], num_keys: Sequence[int]):
        T = len(num_keys)
        kp = (C.c_void_p * T)(*[C.c_void_p(p) for p in h_keys_ptrs])
        vp = (C.c_void_p * T)(*[C.c_void_p(p) for p in vec_ptrs])
        nk = (C.c_size_t * T)(*[int(n) for n in num_keys])
        _check(LIB.hps_session_lookup(self._h, kp, vp, nk, T))

    @staticmethod
    def pack_ptrs(ptrs: Sequence[int]):
        """ctypes array of pointers, built once for calls repeated with the same buffers (lookup_packed)."""
        return (C.c_void_p * len(ptrs))(*[C.c_void_p(p) for p in ptrs])

    @staticmethod
    def pack_counts(num_keys: Sequence[int]):
        return (C.c_size_t * len(num_keys))(*[int(n) for n in num_keys])

    def lookup_packed(self, key_ptrs, vec_ptrs, counts):
        """hps_session_lookup with arguments prepared by pack_ptrs / pack_counts (no per-call marshalling)."""
        _check(LIB.hps_session_lookup(self._h, key_ptrs, vec_ptrs, counts, len(counts)))

    def lookup_device_ptrs(self, d_keys_ptr: int, vec_ptrs: Sequence[int], num_keys: Sequence[int]):
        T = len(num_keys)
        vp = (C.c_void_p * T)(*[C.c_void_p(p) for p in vec_ptrs])
        nk = (C.c_size_t * T)(*[int(n) for n in num_keys])
        _check(LIB.hps_session_lookup_device(self._h, C.c_void_p(d_keys_ptr), vp, nk, T))

    # -- conveniences --------------------------------------------------------------------------------
    def _slices(self, num_keys):
        koff, ooff, ko, oo = [], [], 0, 0
        for n, d in zip(num_keys, self.dims):
            koff.append(ko)
            ooff.append(oo)
            ko += int(n)
            oo += int(n) * d
        return koff, ooff, ko, oo

    def lookup(self, keys, num_keys, out=None):
        """KEYS flat table-major int64 (host), NUMKEYS per table → OUTPUT0 flat fp32.

        gpucache=true: ``out`` is a torch CUDA tensor (allocated if None); gpucache=false: numpy array.
        Pointer slicing = ModelInstanceState::ProcessRequest (model_instance_state.cpp:180-193)."""
        keys = np.ascontiguousarray(keys, dtype=np.int64).ravel()
        num_keys = [int(n) for n in num_keys]
        koff, ooff, nk, no = self._slices(num_keys)
        if nk != keys.size:
            raise HpsError(ERR_INVALID_ARG, f"sum(NUMKEYS)={nk} != len(KEYS)={keys.size}")
        kptrs = [keys.ctypes.data + 8 * o for o in koff]
        if self.use_gpu_cache:
            import torch
            if out is None:
                out = torch.empty(no, dtype=torch.float32, device=torch.device("cuda", self.device))
            assert out.is_cuda and out.dtype == torch.float32 and out.numel() >= no and out.is_contiguous()
            # the engine writes on its own stream: whatever the caller's stream still does to `out` must be over
            torch.cuda.current_stream(out.device).synchronize()
            base = out.data_ptr()
        else:
            if out is None:
                out = np.empty(no, dtype=np.float32)
            assert out.dtype == np.float32 and out.size >= no
            base = out.ctypes.data
        self.lookup_ptrs(kptrs, [base + 4 * o for o in ooff], num_keys)
        return out

    def lookup_device(self, d_keys, num_keys, out=None):
        """Same with KEYS already on the device (torch int64 CUDA tensor, flat table-major)."""
        import torch
        num_keys = [int(n) for n in num_keys]
        _, ooff, nk, no = self._slices(num_keys)
        assert d_keys.is_cuda and d_keys.dtype == torch.int64 and d_keys.numel() == nk and d_keys.is_contiguous()
        if out is None:
            out = torch.empty(no, dtype=torch.float32, device=d_keys.device)
        assert out.is_cuda and out.is_contiguous() and out.device == d_keys.device
        # The engine reads d_keys and writes `out` on the session's own non-blocking stream: the torch stream that
        # produced the keys (or still uses `out`) has to be done first.
        torch.cuda.current_stream(d_keys.device).synchronize()
        base = out.data_ptr()
        self.lookup_device_ptrs(d_keys.data_ptr(), [base + 4 * o for o in ooff], num_keys)
        return out

    def last_stats(self) -> LookupStats:
        s = LookupStats()
        _check(LIB.hps_session_last_stats(self._h, C.byref(s)))
        return s

    def set_option(self, name: str, value: int):
        _check(LIB.hps_session_set_option(self._h, name.encode(), int(value)))

    def close(self):
        if self._h:
            LIB.hps_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_shard_passes(counts, capacity: int):
    """[(offset, [n_0 .. n_{T-1}]), ...]: the lookup calls that serve one owner's bucket (hps_shard_plan_passes; host logic)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    T = counts.size
    need = int(LIB.hps_shard_plan_passes(counts.ctypes.data, T, int(capacity), None, 0))
    out = np.zeros((max(need, 1), 1 + T), dtype=np.uint64)
    got = int(LIB.hps_shard_plan_passes(counts.ctypes.data, T, int(capacity), out.ctypes.data, need))
    assert got == need
    return [(int(r[0]), [int(v) for v in r[1:]]) for r in out[:need]]


class ShardedEntrySession:
    """Entry session of a table-sharded model (include/hps_amd.h: hps_shard_entry_*): what one Triton instance of such a
    model is — whole requests in, rows from every shard written into the entry device's output."""

    def __init__(self, handle, server: HierParameterServer, model: str, device: int):
        self._h = handle
        self._server = server
        self.model = model
        self.device = int(device)
        mi = server.model_info(model)
        self.num_tables = int(mi.num_tables)
        self.dims = [int(server.table_info(model, t).embedding_vecsize) for t in range(self.num_tables)]

    @classmethod
    def create(cls, server: HierParameterServer, model: str, entry_device: int = 0) -> "ShardedEntrySession":
        h = C.c_void_p()
        _check(LIB.hps_shard_entry_create(server._h, model.encode(), int(entry_device), C.byref(h)))
        return cls(h, server, model, entry_device)

    _slices = LookupSession._slices

    def lookup(self, keys, num_keys, out=None):
        """KEYS flat table-major int64 in host memory -> OUTPUT0 (torch CUDA tensor on the entry device)."""
        import torch
        keys = np.ascontiguousarray(keys, dtype=np.int64).ravel()
        num_keys = [int(n) for n in num_keys]
        koff, ooff, nk, no = self._slices(num_keys)
        if nk != keys.size:
            raise HpsError(ERR_INVALID_ARG, f"sum(NUMKEYS)={nk} != len(KEYS)={keys.size}")
        if out is None:
            out = torch.empty(no, dtype=torch.float32, device=torch.device("cuda", self.device))
        assert out.is_cuda and out.dtype == torch.float32 and out.numel() >= no and out.is_contiguous()
        torch.cuda.current_stream(out.device).synchronize()
        T = len(num_keys)
        kp = (C.c_void_p * T)(*[C.c_void_p(keys.ctypes.data + 8 * o) for o in koff])
        vp = (C.c_void_p * T)(*[C.c_void_p(out.data_ptr() + 4 * o) for o in ooff])
        nkc = (C.c_size_t * T)(*num_keys)
        _check(LIB.hps_shard_entry_lookup(self._h, kp, vp, nkc, T))
        return out

    def lookup_device(self, d_keys, num_keys, out=None):
        import torch
        num_keys = [int(n) for n in num_keys]
        _, ooff, nk, no = self._slices(num_keys)
        assert d_keys.is_cuda and d_keys.dtype == torch.int64 and d_keys.numel() == nk and d_keys.is_contiguous()
        if out is None:
            out = torch.empty(no, dtype=torch.float32, device=d_keys.device)
        torch.cuda.current_stream(d_keys.device).synchronize()
        T = len(num_keys)
        vp = (C.c_void_p * T)(*[C.c_void_p(out.data_ptr() + 4 * o) for o in ooff])
        nkc = (C.c_size_t * T)(*num_keys)
        _check(LIB.hps_shard_entry_lookup_device(self._h, C.c_void_p(d_keys.data_ptr()), vp, nkc, T))
        return out

    def last_stats(self) -> ShardEntryStats:
        s = ShardEntryStats()
        _check(LIB.hps_shard_entry_last_stats(self._h, C.byref(s)))
        return s

    def set_option(self, name: str, value: int):
        _check(LIB.hps_shard_entry_set_option(self._h, name.encode(), int(value)))

    @property
    def shard_capacity(self) -> int:
        return int(LIB.hps_shard_entry_shard_capacity(self._h))

    def close(self):
        if self._h:
            LIB.hps_shard_entry_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_update_message(model: str, table: int, keys, rows) -> bytes:
    """One frame of the online update source's message file (csrc/ps/update_source.h): what a producer appends."""
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    dim = rows.size // max(keys.size, 1)
    n = C.c_uint64(0)
    _check(LIB.hps_update_message_encode(model.encode(), table, dim, keys.ctypes.data, rows.ctypes.data, keys.size, None, 0, C.byref(n)))
    buf = (C.c_uint8 * n.value)()
    _check(LIB.hps_update_message_encode(model.encode(), table, dim, keys.ctypes.data, rows.ctypes.data, keys.size, buf, n.value, C.byref(n)))
    return bytes(buf)
