"""CPU oracle for the HPS lookup path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  It is the checker, never the thing measured as the product or shipped.

PARITY UNPINNED (see the header of ``hps_oracle.c`` and DESIGN.md §Oracle): the reference's arithmetic
lives in the un-vendored NVIDIA/HugeCTR ``libhuge_ctr_hps.so``; /root/reference holds no numeric golden
vectors for the path.  The oracle is pinned only on the structural facts the reference states
(file format, request layout, output shape, default fill, response parameters).

Two independent restatements live here:
  * ``COracle`` — ctypes binding of ``hps_oracle.c`` (hash-map find per key, the timed CPU baseline);
  * ``np_*``    — NumPy restatement (sort + searchsorted; no hashing at all), used to cross-check the C
                  oracle and to generate the fixtures in tests/golden/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "libhps_oracle.so"

SEED = 20260929
_M64 = (1 << 64) - 1


def build(force: bool = False) -> Path:
    """Compile hps_oracle.c with gcc (make -C oracle)."""
    src = _HERE / "hps_oracle.c"
    if force or not _LIB.exists() or _LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _LIB


# --------------------------------------------------------------------------------------------------
# NumPy restatement
# --------------------------------------------------------------------------------------------------
def np_mix64(x):
    """splitmix64 finalizer on uint64 arrays (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def np_synth_rows(seed: int, t: int, keys, D: int) -> np.ndarray:
    """row(t,k)[j] of the synthetic table recipe: fp32 in [0.5,1) with hashed mantissa."""
    keys = np.asarray(keys, dtype=np.int64)
    with np.errstate(over="ignore"):
        tb = np_mix64(np.uint64(seed) ^ np_mix64(np.uint64(t + 1)))
        rb = np_mix64(tb + keys.astype(np.uint64))  # [R]
        j = np.arange(D, dtype=np.uint64)
        w = np_mix64(rb[:, None] + (j >> np.uint64(1))[None, :])  # [R, D]
    lo = (w & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (w >> np.uint64(32)).astype(np.uint32)
    m = np.where((j & np.uint64(1)).astype(bool)[None, :], hi, lo)
    bits = np.uint32(0x3F000000) | (m & np.uint32(0x007FFFFF))
    return bits.view(np.float32)


def np_write_table(dirpath, keys, rows) -> None:
    """<dir>/key = int64[R], <dir>/emb_vector = fp32[R,D], native endian, same order
    (/root/reference/docs/architecture.md:185-218; 01_model_training.ipynb:498-504)."""
    d = Path(dirpath)
    d.mkdir(parents=True, exist_ok=True)
    np.asarray(keys, dtype=np.int64).tofile(d / "key")
    np.asarray(rows, dtype=np.float32).tofile(d / "emb_vector")


def np_read_table(dirpath, D: int):
    d = Path(dirpath)
    keys = np.fromfile(d / "key", dtype=np.int64)
    rows = np.fromfile(d / "emb_vector", dtype=np.float32)
    if rows.size != keys.size * D:
        raise ValueError(f"{d}: emb_vector has {rows.size} floats, expected {keys.size}*{D}")
    return keys, rows.reshape(keys.size, D)


def np_find(table_keys, query) -> np.ndarray:
    """Row index of each query key in the table (last duplicate wins), -1 when absent."""
    table_keys = np.asarray(table_keys, dtype=np.int64)
    query = np.asarray(query, dtype=np.int64)
    if table_keys.size == 0:
        return np.full(query.shape, -1, dtype=np.int64)
    order = np.argsort(table_keys, kind="stable")
    sk = table_keys[order]
    pos = np.searchsorted(sk, query, side="right") - 1  # last occurrence
    ok = (pos >= 0) & (sk[np.clip(pos, 0, sk.size - 1)] == query)
    return np.where(ok, order[np.clip(pos, 0, sk.size - 1)], -1)


def np_lookup(tables, keys, num_keys, defaults, resident=None) -> np.ndarray:
    """The whole request (hps.cc:573-630 + model_instance_state.cpp:177-197).

    tables  : list of (table_keys[R], rows[R,D])
    keys    : flat int64, table-major        num_keys: T ints         defaults: T floats
    resident: optional list of per-table int64 arrays = keys currently in the GPU cache.  When given,
              models the ASYNC-insert return (docs/architecture.md:32,65-67): a key that is not
              resident returns the default vector even if the parameter server holds it.
    """
    keys = np.asarray(keys, dtype=np.int64).ravel()
    out, off = [], 0
    for t, ((tk, rows), n) in enumerate(zip(tables, num_keys)):
        q = keys[off:off + n]
        off += n
        D = rows.shape[1]
        idx = np_find(tk, q)
        found = idx >= 0
        if resident is not None and resident[t] is not None:   # None for a table = that table is served synchronously
            found &= np.isin(q, np.asarray(resident[t], dtype=np.int64))
        o = np.full((n, D), np.float32(defaults[t]), dtype=np.float32)
        o[found] = rows[idx[found]]
        out.append(o.ravel())
    if off != keys.size:
        raise ValueError("sum(NUMKEYS) != len(KEYS)")
    return np.concatenate(out) if out else np.zeros(0, np.float32)


def np_unique_counts(keys, num_keys, resident) -> list:
    """Per table: (unique keys of the call, unique keys of the call that are not resident).  The engine "first
    determines the associated unique embedding keys" of a batch and resolves those against the cache
    (docs/hierarchical_parameter_server.md:69); both lengths drive the insertion policy below."""
    keys = np.asarray(keys, dtype=np.int64).ravel()
    out, off = [], 0
    for t, n in enumerate(num_keys):
        u = np.unique(keys[off:off + n])
        off += n
        out.append((int(u.size), int((~np.isin(u, np.asarray(resident[t], dtype=np.int64))).sum())))
    return out


def np_insert_modes(keys, num_keys, resident, hit_rate_threshold) -> list:
    """Insertion policy of one call, per table (docs/architecture.md:65-67): True = async
    (the table's hit rate in this call >= hit_rate_threshold: misses return the default vector and are inserted in
    the background), False = synchronous (missed rows are fetched, returned exactly and inserted before the call
    returns).  The reference loops over the tables of a request and decides for each; its hit rate is "the real hit
    rate of the GPU embedding cache lookup" (docs/architecture.md:66) — and what is looked up in the cache are the
    batch's UNIQUE keys (docs/hierarchical_parameter_server.md:69).  So:

        hit rate(t) = 1 - (unique keys of table t missing from the cache) / (unique keys of table t)

    (SURVEY.md Appendix C4).  Round 1 restated this over the keys as sent, following the product of the time, which
    deduplicated only the misses; that was a deviation without a reference citation and is corrected here.  Tables
    without misses or without keys are synchronous by definition."""
    thr = float(np.float32(hit_rate_threshold))   # the configuration carries the threshold as a float (backend.cpp:372)
    modes = []
    for uniq, uniq_missing in np_unique_counts(keys, num_keys, resident):
        modes.append(uniq_missing > 0 and 1.0 - uniq_missing / uniq >= thr)
    return modes


def np_insert_modes_keys_as_sent(keys, num_keys, resident, hit_rate_threshold) -> list:
    """The round-1 definition (hit rate over the keys as sent, duplicates counted) — kept ONLY so that a test can show
    the two definitions disagree on skewed batches and that the product follows np_insert_modes, not this."""
    keys = np.asarray(keys, dtype=np.int64).ravel()
    thr = float(np.float32(hit_rate_threshold))
    modes, off = [], 0
    for t, n in enumerate(num_keys):
        q = keys[off:off + n]
        off += n
        if n == 0:
            modes.append(False)
            continue
        misses = int((~np.isin(q, np.asarray(resident[t], dtype=np.int64))).sum())
        modes.append(misses > 0 and 1.0 - misses / n >= thr)
    return modes


class VolatileDbModel:
    """Bounded CPU-memory database in front of a persistent one — the host tier when RAM < table
    (docs/hierarchical_parameter_server.md:460-507): at most `overflow_margin` embeddings per partition (partition =
    key mod num_partitions, docs/architecture.md:131); inserting one more prunes the partition to
    `overflow_margin * overflow_resolution_target` by `overflow_policy`; `initial_cache_rate` of the table (first rows
    in file order) is cached at start-up; `cache_missed_embeddings` inserts what had to be read from behind the tier.
    Pure-Python, single-threaded, small cases only.  Where the documentation is silent this model decides, and the
    product follows it (csrc/ps/volatile_tier.h): keep = max(1, floor(margin * target)), and a full partition whose
    prune frees nothing evicts one entry; evict_oldest orders by
    (last access, key), evict_least_used by (access count, last access, key), smallest evicted first; one access
    stamp per fetch call; an insert counts as one access.  evict_random is not modelled beyond its sizes."""

    def __init__(self, table_keys, num_partitions=8, overflow_margin=None, overflow_policy="evict_random",
                 overflow_resolution_target=0.8, initial_cache_rate=1.0, cache_missed_embeddings=False, persistent=True):
        keys = np.asarray(table_keys, dtype=np.int64)
        self.P = int(num_partitions)
        self.policy = overflow_policy
        self.cache_missed = bool(cache_missed_embeddings)
        self.persistent = bool(persistent)
        self.live = {}                       # key -> row number of its last occurrence (duplicate keys: last wins)
        for r, k in enumerate(keys.tolist()):
            self.live[k] = r
        # the margin is the limit whatever the table held at load time (keys added later count against it too)
        margin = int(overflow_margin) if overflow_margin is not None else None
        t = overflow_resolution_target if 0.0 < overflow_resolution_target < 1.0 else 0.8
        self.cap = [max(1, margin) if margin is not None else float("inf")] * self.P
        self.keep = [max(1, int(np.floor(max(1, margin) * t))) if margin is not None else float("inf")] * self.P
        self.part = [dict() for _ in range(self.P)]      # key -> [stamp, count]
        self.clock = 0
        self.evictions = 0
        first = int(np.ceil(min(max(initial_cache_rate, 0.0), 1.0) * keys.size))
        for r in range(first):
            k = int(keys[r])
            if self.live[k] == r:
                self._insert(k, 0)

    def _p(self, k):
        return (k & 0xFFFFFFFFFFFFFFFF) % self.P   # the key's bits as unsigned, like the product

    def _insert(self, k, now):
        p = self._p(k)
        part = self.part[p]
        if k not in part:
            if len(part) >= self.cap[p]:
                if self.policy == "evict_random":
                    raise NotImplementedError("evict_random is only checked through sizes")
                rank = (lambda kv: (kv[1][1], kv[1][0], kv[0])) if self.policy == "evict_least_used" else \
                       (lambda kv: (kv[1][0], kv[0]))
                victims = sorted(part.items(), key=rank)[:max(len(part) - self.keep[p], 1)]
                for v, _ in victims:
                    del part[v]
                self.evictions += len(victims)
            part[k] = [now, 0]
        part[k][0] = now
        part[k][1] += 1

    def fetch(self, keys):
        """One fetch call: returns found[i]; updates the access statistics and (cache_missed_embeddings) the content."""
        self.clock += 1
        now = self.clock
        found, missed = [], []
        for k in np.asarray(keys, dtype=np.int64).tolist():
            part = self.part[self._p(k)]
            if k in part:
                part[k][0] = now
                part[k][1] += 1
                found.append(True)
            elif self.persistent and k in self.live:
                found.append(True)
                if self.cache_missed:
                    missed.append(k)
            else:
                found.append(False)
        for k in missed:
            self._insert(k, now)
        return np.array(found, dtype=bool)

    def resident(self) -> np.ndarray:
        return np.array(sorted(k for part in self.part for k in part), dtype=np.int64)


# --------------------------------------------------------------------------------------------------
# C oracle binding
# --------------------------------------------------------------------------------------------------
class COracle:
    """ctypes view of hps_oracle.c.  One instance per model (list of tables)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(str(build()))
            L.oracle_mix64.restype = C.c_uint64
            L.oracle_mix64.argtypes = [C.c_uint64]
            L.oracle_synth_rows.restype = None
            L.oracle_synth_rows.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, C.c_uint32, C.c_void_p]
            L.oracle_table_from_arrays.restype = C.c_void_p
            L.oracle_table_from_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32]
            L.oracle_table_load.restype = C.c_void_p
            L.oracle_table_load.argtypes = [C.c_char_p, C.c_uint32]
            L.oracle_table_free.restype = None
            L.oracle_table_free.argtypes = [C.c_void_p]
            L.oracle_table_rows.restype = C.c_int64
            L.oracle_table_rows.argtypes = [C.c_void_p]
            L.oracle_table_find.restype = C.c_int64
            L.oracle_table_find.argtypes = [C.c_void_p, C.c_int64]
            L.oracle_output_elems.restype = C.c_int64
            L.oracle_output_elems.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            L.oracle_lookup.restype = C.c_int64
            L.oracle_lookup.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.oracle_lookup_mt.restype = C.c_int64
            L.oracle_lookup_mt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            L.oracle_write_table.restype = C.c_int
            L.oracle_write_table.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32]
            cls._lib = L
        return cls._lib

    def __init__(self):
        self.L = self.lib()
        self._tables = []  # (handle, D)
        self._keep = []    # numpy arrays borrowed by the C side

    def add_table_arrays(self, keys, rows):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        R, D = rows.shape
        assert keys.shape == (R,)
        h = self.L.oracle_table_from_arrays(keys.ctypes.data, rows.ctypes.data, R, D)
        if not h:
            raise MemoryError("oracle_table_from_arrays")
        self._tables.append((h, D))
        self._keep.append((keys, rows))

    def reserve(self, T: int):
        """T empty table positions, to be filled (from several threads if wanted) by add_table_arrays_borrowed."""
        self._tables = [None] * T
        self._keep = [None] * T

    def add_table_arrays_borrowed(self, t: int, keys, rows):
        """Index table t over arrays the CALLER keeps alive (int64[R], float32[R, D], C-contiguous): no copy of the rows.
        The index is the oracle's own (hps_oracle.c: table_index)."""
        assert keys.dtype == np.int64 and rows.dtype == np.float32 and keys.flags.c_contiguous and rows.flags.c_contiguous
        R, D = rows.shape
        assert keys.shape == (R,)
        h = self.L.oracle_table_from_arrays(keys.ctypes.data, rows.ctypes.data, R, D)
        if not h:
            raise MemoryError("oracle_table_from_arrays")
        self._tables[t] = (h, D)
        self._keep[t] = (keys, rows)

    def add_table_dir(self, dirpath, D):
        h = self.L.oracle_table_load(os.fsencode(str(dirpath)), D)
        if not h:
            raise OSError(f"oracle_table_load({dirpath}, D={D}) failed")
        self._tables.append((h, D))

    @property
    def dims(self):
        return [d for _, d in self._tables]

    def _handles(self):
        return (C.c_void_p * len(self._tables))(*[h for h, _ in self._tables])

    def lookup(self, keys, num_keys, defaults, threads: int = 1, out=None) -> np.ndarray:
        keys = np.ascontiguousarray(keys, dtype=np.int64).ravel()
        nk = np.ascontiguousarray(num_keys, dtype=np.int32).ravel()
        df = np.ascontiguousarray(defaults, dtype=np.float32).ravel()
        T = len(self._tables)
        assert nk.size == T and df.size == T and int(nk.sum()) == keys.size
        hs = self._handles()
        n = self.L.oracle_output_elems(hs, nk.ctypes.data, T)
        if out is None:
            out = np.empty(n, dtype=np.float32)
        assert out.size == n and out.dtype == np.float32
        w = self.L.oracle_lookup_mt(hs, T, keys.ctypes.data, nk.ctypes.data, df.ctypes.data, out.ctypes.data, threads)
        assert w == n
        return out

    def close(self):
        for h, _ in self._tables:
            self.L.oracle_table_free(h)
        self._tables, self._keep = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def c_synth_rows(seed: int, t: int, key0: int, count: int, D: int, out=None) -> np.ndarray:
    L = COracle.lib()
    if out is None:
        out = np.empty((count, D), dtype=np.float32)
    L.oracle_synth_rows(seed, t, key0, count, D, out.ctypes.data)
    return out
