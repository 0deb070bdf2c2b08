"""Torch-free exercise of the GPU lookup paths, for runs with the ASan runtime preloaded (torch's CUDA initialisation does
not survive LD_PRELOAD=libclang_rt.asan).  Device buffers come from hipMalloc through ctypes; everything else is the
product's C ABI.  Both parameter-server tiers, synchronous / mixed / async policy, two sessions on two threads, a table
reload and a cache refresh in between; every row compared with the oracle.

    HPS_AMD_LIB_DIR=<asan build> LD_PRELOAD=<libclang_rt.asan-x86_64.so> python tests/tools/asan_gpu_run.py
"""
import ctypes as C
import sys
import threading
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))

HIP = C.CDLL("/opt/rocm/lib/libamdhip64.so")
HIP.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
HIP.hipFree.argtypes = [C.c_void_p]


def dmalloc(nbytes):
    p = C.c_void_p()
    assert HIP.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
    return p


def main():
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    from tests.conftest import make_tables, ps_config
    assert hps.device_count() > 0
    shapes = [(6000, 128), (4000, 16), (900, 3)]
    tables = make_tables(shapes)
    T = len(tables)
    dims = [d for _, d in shapes]
    failures = []
    for direct in (False, True):
        for thr in (1.0, 0.8):
            name = f"asan_{int(direct)}_{int(thr * 10)}"
            cfg = ps_config(name, tables, maxcat=[1] * T, defaults=[0.5, -1.0, 2.0], gpucacheper=0.2, max_batch=4096,
                            hit_rate_threshold=thr, extra={"ps_direct_access": direct})
            ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
            for t, (k, r) in enumerate(tables):
                ps.load_table_arrays(name, t, k, r)
            ps.create_embedding_cache_per_model(name)
            cache = ps.get_embedding_cache(name, 0)
            sessions = [hps.LookupSession.create(ps, name, cache) for _ in range(2)]

            def worker(si, seed):
                rng = np.random.default_rng(seed)
                s = sessions[si]
                try:
                    for it in range(12):
                        nk = [int(rng.integers(0, 4097)) for _ in range(T)]
                        parts = []
                        for (keys, _), n in zip(tables, nk):
                            q = rng.choice(keys[: keys.size // 3], n) if it % 2 else rng.choice(keys, n)
                            q = np.where(rng.random(n) < 0.05, -1 - rng.integers(0, 1 << 40, n), q)
                            parts.append(q.astype(np.int64))
                        q = np.concatenate(parts)
                        n_out = sum(n * d for n, d in zip(nk, dims))
                        d_out = dmalloc(n_out * 4)
                        offs, ptrs, off = [], [], 0
                        for n, d in zip(nk, dims):
                            ptrs.append((d_out.value or 0) + off * 4)
                            off += n * d
                        kptrs, koff = [], 0
                        for n in nk:
                            kptrs.append(q.ctypes.data + koff * 8)
                            koff += n
                        kp = (C.c_void_p * T)(*kptrs)
                        vp = (C.c_void_p * T)(*ptrs)
                        nkc = (C.c_size_t * T)(*nk)
                        hps._check(hps.LIB.hps_session_lookup(s._h, kp, vp, nkc, T))
                        out = np.empty(n_out, np.float32)
                        if n_out:
                            assert HIP.hipMemcpy(out.ctypes.data, d_out, n_out * 4, 2) == 0
                        HIP.hipFree(d_out)
                        ref_sync = O.np_lookup(tables, q, nk, [0.5, -1.0, 2.0])
                        same = out.view(np.uint32) == ref_sync.view(np.uint32)
                        if thr < 1.0:   # async tables may answer the default for keys not resident yet
                            dflt = np.concatenate([np.full(n * d, v, np.float32) for n, d, v in zip(nk, dims, [0.5, -1.0, 2.0])])
                            same |= out.view(np.uint32) == dflt.view(np.uint32)
                        if not same.all():
                            failures.append((name, si, it, int((~same).sum())))
                            return
                        if si == 0 and it == 5:
                            ps.load_table_arrays(name, 1, *tables[1])      # reload under the other session
                            st_r = ps.refresh_embedding_cache(name, 0)     # (only what can differ: the reloaded table, in full)
                            assert st_r["tables_full"] == 1 and st_r["tables_unchanged"] == T - 1, st_r
                        if si == 0 and it == 8:
                            ps.upsert(name, 0, tables[0][0][:300], tables[0][1][:300])    # an online update (same rows): change log
                            ps.refresh_embedding_cache(name, 0)
                            ps.refresh_embedding_cache(name, 0, full=True)                # the reference's full pass, paced
                except Exception as e:  # noqa: BLE001
                    failures.append((name, si, repr(e)))

            th = [threading.Thread(target=worker, args=(i, 10 * i + int(direct))) for i in range(2)]
            [x.start() for x in th]
            [x.join() for x in th]
            cache.wait_async()
            for s in sessions:
                s.close()
            print(f"{name}: direct={direct} threshold={thr}: {cache.counters()}")
            cache.release()
            ps.close()
    # ---- a table-sharded model served by entry sessions (csrc/cache/shard_entry.cpp: worker threads, bucket kernels, indexed
    #      gather): 3 logical shards, two entry sessions on two threads, uniform and skewed requests, one owner served in passes ----
    name = "asan_sharded"
    cfg = ps_config(name, tables, maxcat=[2, 1, 1], defaults=[0.5, -1.0, 2.0], gpucacheper=0.3, max_batch=2048,
                    extra={"table_sharding": "hash", "shard_capacity_factor": 1.0})
    cfg["models"][0]["deployed_device_list"] = [0, 0, 0]
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays(name, t, k, r)
    ps.create_embedding_cache_per_model(name)
    entries = []
    for _ in range(2):
        h = C.c_void_p()
        hps._check(hps.LIB.hps_shard_entry_create(ps._h, name.encode(), 0, C.byref(h)))
        entries.append(h)

    # (round 6) the second entry session moves its rows by staged copies, in small pieces (csrc/cache/shard_entry.cpp ServeStaged)
    hps._check(hps.LIB.hps_shard_entry_set_option(entries[1], b"transport", 1))
    hps._check(hps.LIB.hps_shard_entry_set_option(entries[1], b"copy_piece_keys", 1024))

    def eworker(ei, seed):
        rng = np.random.default_rng(seed)
        try:
            for it in range(10):
                nk = [int(rng.integers(0, 4097)), int(rng.integers(0, 2049)), int(rng.integers(0, 2049))]
                parts = []
                for (keys, _), n in zip(tables, nk):
                    q = keys[np.minimum(rng.zipf(1.3, n) - 1, keys.size - 1)] if it % 2 else rng.choice(keys, n)
                    q = np.where(rng.random(n) < 0.05, -1 - rng.integers(0, 1 << 40, n), q)
                    parts.append(q.astype(np.int64))
                q = np.concatenate(parts)
                n_out = sum(n * d for n, d in zip(nk, dims))
                d_out = dmalloc(n_out * 4)
                ptrs, off = [], 0
                for n, d in zip(nk, dims):
                    ptrs.append((d_out.value or 0) + off * 4)
                    off += n * d
                kptrs, koff = [], 0
                for n in nk:
                    kptrs.append(q.ctypes.data + koff * 8)
                    koff += n
                kp = (C.c_void_p * T)(*kptrs)
                vp = (C.c_void_p * T)(*ptrs)
                nkc = (C.c_size_t * T)(*nk)
                hps._check(hps.LIB.hps_shard_entry_lookup(entries[ei], kp, vp, nkc, T))
                out = np.empty(n_out, np.float32)
                if n_out:
                    assert HIP.hipMemcpy(out.ctypes.data, d_out, n_out * 4, 2) == 0
                HIP.hipFree(d_out)
                ref = O.np_lookup(tables, q, nk, [0.5, -1.0, 2.0])
                if not np.array_equal(out.view(np.uint32), ref.view(np.uint32)):
                    failures.append((name, ei, it))
                    return
        except Exception as e:  # noqa: BLE001
            failures.append((name, ei, repr(e)))

    th = [threading.Thread(target=eworker, args=(i, 77 + i)) for i in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    for h in entries:
        hps.LIB.hps_shard_entry_destroy(h)
    print(f"{name}: 3 logical shards, two entry sessions: {[ps.get_shard_cache(name, s).counters()['keys'] for s in range(3)]} keys per shard")
    ps.close()
    # ---- big requests on a sharded model: host keys staged narrow (3-byte / uint32 offsets, the widen kernel), the fail-over to 8
    #      bytes, and the adaptive input dedup (tile level only after a request that repeated little) ----
    name = "asan_sharded_big"
    rng = np.random.default_rng(5)
    kb = (5_000_000_000 + rng.permutation(1 << 21)[:300000]).astype(np.int64)
    kc = (3 + rng.permutation(1 << 29)[:200000].astype(np.int64) * 5).astype(np.int64)
    big = [(kb, rng.standard_normal((kb.size, 8), dtype=np.float32)), (kc, rng.standard_normal((kc.size, 4), dtype=np.float32))]
    bdims = [8, 4]
    cfg = ps_config(name, big, maxcat=[1, 1], defaults=[0.0, 9.0], gpucacheper=0.5, max_batch=100000, extra={"table_sharding": "hash"})
    cfg["models"][0]["deployed_device_list"] = [0, 0]
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t, (k, r) in enumerate(big):
        ps.load_table_arrays(name, t, k, r)
    ps.create_embedding_cache_per_model(name)
    h = C.c_void_p()
    hps._check(hps.LIB.hps_shard_entry_create(ps._h, name.encode(), 0, C.byref(h)))
    seen = []
    for it in range(7):
        nk = [100000, 60000] if it != 1 else [150000, 0]
        if it in (5,):     # skewed: repeats inside tiles
            parts = [k[np.minimum(rng.zipf(1.2, n) - 1, k.size - 1)] for (k, _), n in zip(big, nk)]
        else:              # no repeats
            parts = [rng.permutation(k)[:n] for (k, _), n in zip(big, nk)]
        q = np.concatenate(parts).astype(np.int64)
        if it == 3:
            q[70001] = -17                       # outside every frame: the request is restaged at 8 bytes
        n_out = sum(n * d for n, d in zip(nk, bdims))
        d_out = dmalloc(n_out * 4)
        kp = (C.c_void_p * 2)(q.ctypes.data, q.ctypes.data + nk[0] * 8)
        vp = (C.c_void_p * 2)(d_out.value, (d_out.value or 0) + nk[0] * bdims[0] * 4)
        nkc = (C.c_size_t * 2)(*nk)
        hps._check(hps.LIB.hps_shard_entry_lookup(h, kp, vp, nkc, 2))
        st = hps.ShardEntryStats()
        hps._check(hps.LIB.hps_shard_entry_last_stats(h, C.byref(st)))
        seen.append((int(st.key_bytes), int(st.dedup_level)))
        out = np.empty(n_out, np.float32)
        assert HIP.hipMemcpy(out.ctypes.data, d_out, n_out * 4, 2) == 0
        HIP.hipFree(d_out)
        if not np.array_equal(out.view(np.uint32), O.np_lookup(big, q, nk, [0.0, 9.0]).view(np.uint32)):
            failures.append((name, it))
    hps.LIB.hps_shard_entry_destroy(h)
    ps.close()
    print(f"{name}: (key bytes over PCIe, dedup level) per request: {seen}")
    if seen != [(4, 2), (3, 1), (4, 1), (8, 1), (8, 1), (8, 1), (8, 2)]:
        failures.append((name, "widths / dedup levels", seen))
    print("asan_gpu_run:", "FAILED " + repr(failures[:3]) if failures else "ok")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
