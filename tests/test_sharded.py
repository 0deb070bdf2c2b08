"""Table-sharded lookup (BASELINE config 3): owner(key) = mix64(key) mod P, all-to-all of keys and rows.

CPU: two gloo ranks, each serving its shard from a gpucache=false session (host tier) — runs without a GPU.
GPU: the device bucket / unpermute kernels against NumPy, and two ranks sharing the one visible GPU (gloo
carries the exchange through host memory there; on a multi-GPU node the same code runs over RCCL).
"""
import os
import socket

import numpy as np
import pytest

from tests.conftest import make_tables, ps_config


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_owner_function_matches_the_c_abi():
    from hugectr_backend_amd import hps, sharded
    rng = np.random.default_rng(0)
    keys = np.concatenate([rng.integers(-2**62, 2**62, 2000), [0, 1, -1, np.iinfo(np.int64).min, np.iinfo(np.int64).max]]).astype(np.int64)
    for P in (1, 2, 3, 8):
        ref = np.array([hps.LIB.hps_shard_owner(int(k), P) for k in keys])
        assert np.array_equal(sharded.owner_of(keys, P), ref)
        assert ref.min() >= 0 and ref.max() < P
    # shards are a partition of the table
    ks, rows = keys[:1000], rng.random((1000, 4)).astype(np.float32)
    parts = [sharded.shard_rows(ks, rows, r, 4) for r in range(4)]
    assert sum(p[0].size for p in parts) == 1000
    assert np.array_equal(np.sort(np.concatenate([p[0] for p in parts])), np.sort(ks))


def _rank_main(rank, world, port, device_mode, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HCTR_DEFAULT_CONCURRENCY="2")
        import torch
        import torch.distributed as dist
        from hugectr_backend_amd import hps, sharded
        from oracle import hps_oracle as O
        dist.init_process_group("gloo", rank=rank, world_size=world)
        D = 128 if device_mode else 16
        tables = make_tables([(6000, D)], seed=11)
        keys, rows = tables[0]
        my_k, my_r = sharded.shard_rows(keys, rows, rank, world)
        cfg = ps_config("shard", [(my_k, my_r)], gpucache=device_mode, gpucacheper=0.3, hit_rate_threshold=1.0, defaults=[2.0],
                        maxcat=[1], max_batch=8192)
        ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
        ps.load_table_arrays("shard", 0, my_k, my_r)
        cache = None
        if device_mode:
            ps.create_embedding_cache_per_model("shard")
            cache = ps.get_embedding_cache("shard", 0)
        sess = hps.LookupSession.create(ps, "shard", cache)
        sl = sharded.ShardedLookup(sess)
        rng = np.random.default_rng(100 + rank)
        for it in range(4):
            n = int(rng.integers(1, 5000)) if it else 4096
            q_keys = np.where(rng.random(n) < 0.1, -1 - rng.integers(0, 1 << 40, n), rng.choice(keys, n)).astype(np.int64)
            if device_mode:
                out = sl.lookup(torch.from_numpy(q_keys).cuda()).cpu().numpy()
            else:
                out = sl.lookup(q_keys)
            ref = O.np_lookup(tables, q_keys, [n], [2.0])   # oracle over the WHOLE table
            if not np.array_equal(_bits(out), _bits(ref)):
                q.put((rank, f"mismatch at iteration {it}"))
                return
            if sum(sl.last_sent) != n:
                q.put((rank, "send counts do not add up"))
                return
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "exception: " + repr(e) + traceback.format_exc()[-800:]))


def _run_two_ranks(device_mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, device_mode, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_two_gloo_ranks_host_tier_shards():
    _run_two_ranks(device_mode=False)


@pytest.mark.gpu
def test_two_ranks_sharing_the_gpu_device_path():
    _run_two_ranks(device_mode=True)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 8, 64])
def test_device_bucket_and_unpermute_kernels(P):
    import ctypes as C
    import torch
    from hugectr_backend_amd import hps, sharded
    rng = np.random.default_rng(P)
    for n in (0, 1, 1023, 1024, 1025, 50000, 212992):
        keys = rng.integers(-2**62, 2**62, n).astype(np.int64)
        dk = torch.from_numpy(keys).cuda()
        ks = torch.empty(n, dtype=torch.int64, device="cuda")
        perm = torch.empty(n, dtype=torch.int32, device="cuda")
        tot = torch.empty(P, dtype=torch.int64, device="cuda")
        ws = torch.empty(max(int(hps.LIB.hps_shard_bucket_workspace_bytes(n, P)), 16), dtype=torch.uint8, device="cuda")
        hps._check(hps.LIB.hps_shard_bucket_device(dk.data_ptr(), n, P, ks.data_ptr(), perm.data_ptr(), tot.data_ptr(),
                                                   ws.data_ptr(), C.c_void_p(0)))
        torch.cuda.synchronize()
        own = sharded.owner_of(keys, P)
        ref_perm = np.argsort(own, kind="stable")
        assert np.array_equal(perm.cpu().numpy(), ref_perm.astype(np.int32))       # stable counting sort
        assert np.array_equal(ks.cpu().numpy(), keys[ref_perm])
        assert np.array_equal(tot.cpu().numpy(), np.bincount(own, minlength=P))
        for D in (1, 16, 128):
            rows = rng.random((n, D)).astype(np.float32)
            dr = torch.from_numpy(rows).cuda()
            out = torch.empty(max(n * D, 1), dtype=torch.float32, device="cuda")
            hps._check(hps.LIB.hps_shard_unpermute_device(dr.data_ptr(), perm.data_ptr(), n, D, out.data_ptr(), C.c_void_p(0)))
            torch.cuda.synchronize()
            exp = np.empty((n, D), np.float32)
            exp[ref_perm] = rows
            assert np.array_equal(_bits(out[: n * D].cpu().numpy()), _bits(exp.ravel()))


def _native_shards(P, R=20000, D=128, cache_frac=0.5, max_local=30000, seed=3):
    """P shard servers + sessions of one table in this process (each its own model on device 0)."""
    from hugectr_backend_amd import hps, sharded
    tables = make_tables([(R, D)], seed=seed)
    keys, rows = tables[0]
    made = []
    for r in range(P):
        my_k, my_r = sharded.shard_rows(keys, rows, r, P)
        model = f"nshard{P}_{r}"
        cfg = ps_config(model, [(my_k, my_r)], gpucache=True, gpucacheper=cache_frac, hit_rate_threshold=1.0, defaults=[2.0],
                        maxcat=[1], max_batch=max_local * max(P, 2))
        ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
        ps.load_table_arrays(model, 0, my_k, my_r)
        ps.create_embedding_cache_per_model(model)
        sess = hps.LookupSession.create(ps, model, ps.get_embedding_cache(model, 0))
        made.append((ps, sess))
    return tables, made


@pytest.mark.gpu
@pytest.mark.parametrize("dedup", [1, 0])
@pytest.mark.parametrize("P", [2, 4])
def test_native_sharded_session_logical_shards_in_one_process(P, dedup, monkeypatch):
    """The engine's sharded session (fixed-capacity blocks, positions recorded by the bucket kernel, padded local lookup)
    with P endpoints in one process: device-to-device copies stand in for RCCL, everything else is the production path.
    Uniform keys, absent keys, ragged sizes, an empty request on one rank, and a skewed request that overflows the block
    capacity (one hot key = one owner) and is repeated with twice the capacity."""
    import ctypes as C
    import threading
    import torch
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    monkeypatch.setenv("HPS_SHARD_DEDUP", str(dedup))   # read when the sharded session is created
    tables, made = _native_shards(P)
    keys = tables[0][0]
    max_local = 30000
    grp = C.c_void_p()
    hps._check(hps.LIB.hps_shard_group_create_local(P, C.byref(grp)))
    shards = []
    for r, (_, sess) in enumerate(made):
        h = C.c_void_p()
        hps._check(hps.LIB.hps_shard_session_create_local(sess._h, grp, r, max_local, C.byref(h)))
        shards.append(h)
    rng = np.random.default_rng(P)
    rounds = []
    for it in range(5):
        per_rank = []
        for r in range(P):
            n = max_local if it == 3 else int(rng.integers(1, max_local))
            if it == 1 and r == 0:
                n = 0
            q = np.where(rng.random(n) < 0.1, -1 - rng.integers(0, 1 << 40, n), rng.choice(keys, n)).astype(np.int64)
            if it == 3:   # skew: most of every rank's keys are ONE key -> one owner's block overflows
                q[: int(n * 0.8)] = keys[17]
            per_rank.append(q)
        rounds.append(per_rank)
    errs, attempts = [], [[0] * len(rounds) for _ in range(P)]

    def work(r):
        try:
            torch.cuda.set_device(0)
            for it, per_rank in enumerate(rounds):
                q = per_rank[r]
                dq = torch.from_numpy(q).cuda()
                out = torch.empty(max(q.size, 1) * 128, dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                hps._check(hps.LIB.hps_shard_session_lookup(shards[r], dq.data_ptr(), q.size, out.data_ptr()))
                att, cap = C.c_uint32(0), C.c_uint64(0)
                sent = (C.c_uint64 * P)()
                hps._check(hps.LIB.hps_shard_session_last_stats(shards[r], C.byref(cap), C.byref(att), sent, P))
                attempts[r][it] = att.value
                # with the input dedup only one key of every distinct value travels
                if sum(sent) != (np.unique(q).size if dedup else q.size):
                    errs.append((r, it, f"sent counts do not add up: {sum(sent)} of {q.size} keys, {np.unique(q).size} distinct"))
                ref = O.np_lookup(tables, q, [q.size], [2.0])
                if not np.array_equal(_bits(out[: q.size * 128].cpu().numpy()), _bits(ref)):
                    errs.append((r, it, "mismatch"))
        except Exception as e:  # noqa: BLE001
            errs.append((r, "exception", repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs
    assert all(a[0] == 1 for a in attempts)                 # uniform keys fit the first capacity
    if dedup:
        assert all(a[3] == 1 for a in attempts)             # one hot key is ONE key after the dedup: nothing overflows
    else:
        assert all(a[3] == 2 for a in attempts)             # the skewed round was repeated ONCE, on every rank alike
    assert all(a[4] == 1 for a in attempts)                 # ... and the capacity it found is kept
    for h in shards:
        hps.LIB.hps_shard_session_destroy(h)
    hps.LIB.hps_shard_group_destroy(grp)
    for _, sess in made:
        sess.close()


@pytest.mark.gpu
def test_native_sharded_session_over_rccl_single_rank():
    """The RCCL transport itself (librccl loaded at first use, ncclCommInitRank, grouped ncclSend/ncclRecv on the session's
    stream) with the one GPU of the box: a world of one rank exchanges with itself.  N > 1 needs one GPU per rank."""
    import ctypes as C
    import torch
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    tables, made = _native_shards(1, R=30000, seed=9)
    _, sess = made[0]
    uid = (C.c_uint8 * 128)()
    hps._check(hps.LIB.hps_shard_unique_id(uid))
    h = C.c_void_p()
    hps._check(hps.LIB.hps_shard_session_create(sess._h, 0, 1, uid, 25000, C.byref(h)))
    rng = np.random.default_rng(1)
    for n in (1, 4096, 25000):
        q = np.where(rng.random(n) < 0.05, -3 - rng.integers(0, 1 << 30, n), rng.choice(tables[0][0], n)).astype(np.int64)
        dq = torch.from_numpy(q).cuda()
        out = torch.empty(n * 128, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        hps._check(hps.LIB.hps_shard_session_lookup(h, dq.data_ptr(), n, out.data_ptr()))
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(O.np_lookup(tables, q, [n], [2.0])))
    hps.LIB.hps_shard_session_destroy(h)
    sess.close()


@pytest.mark.gpu
def test_native_sharded_session_zipf_skew_host_keys_and_reserved_key(monkeypatch):
    """P = 4 logical shards, keys in HOST memory (hps_shard_session_lookup_host: staged + narrowed to uint32 when they fit),
    drawn Zipf-like so that a handful of keys make up most of a request: the first call overflows its blocks and is repeated
    ONCE with the capacity the headers reported (never a third attempt); later calls fit.  The cache's reserved key
    (INT64_MIN) never travels and is answered with the default vector; wide keys make the request cross PCIe at 8 bytes."""
    import ctypes as C
    import threading
    import torch
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    monkeypatch.setenv("HPS_SHARD_DEDUP", "0")   # every key as sent: the overflow / exact-retry mechanics (the dedup has its own test below)
    P = 4
    max_local = 40000
    tables, made = _native_shards(P, max_local=max_local)
    keys = tables[0][0]
    grp = C.c_void_p()
    hps._check(hps.LIB.hps_shard_group_create_local(P, C.byref(grp)))
    shards = []
    for r, (_, sess) in enumerate(made):
        h = C.c_void_p()
        hps._check(hps.LIB.hps_shard_session_create_local(sess._h, grp, r, max_local, C.byref(h)))
        shards.append(h)
    # the lookup sessions' handles go first: the sharded sessions keep what they need alive
    for _, sess in made:
        sess.close()
    rng = np.random.default_rng(77)
    ranks = np.argsort(rng.random(keys.size))
    w = 1.0 / np.arange(1, keys.size + 1) ** 1.6          # the hottest key alone is ~45 % of a request
    w /= w.sum()
    rounds = []
    for it in range(4):
        per_rank = []
        for r in range(P):
            q = keys[ranks[rng.choice(keys.size, size=max_local, p=w)]].astype(np.int64)
            if it == 2:
                q[::97] = np.iinfo(np.int64).min            # reserved key
                q[5::101] = -5 - rng.integers(0, 1 << 40)   # absent (and wide: this request goes at 8 bytes per key)
            per_rank.append(q)
        rounds.append(per_rank)
    errs = []
    info = [[None] * len(rounds) for _ in range(P)]

    def work(r):
        try:
            torch.cuda.set_device(0)
            for it, per_rank in enumerate(rounds):
                q = per_rank[r]
                out = torch.empty(q.size * 128, dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                hps._check(hps.LIB.hps_shard_session_lookup_host(shards[r], q.ctypes.data, q.size, out.data_ptr()))
                att, cap = C.c_uint32(0), C.c_uint64(0)
                sent = (C.c_uint64 * P)()
                hps._check(hps.LIB.hps_shard_session_last_stats(shards[r], C.byref(cap), C.byref(att), sent, P))
                t = [C.c_float(0) for _ in range(3)]
                recv, kb = C.c_uint64(0), C.c_int32(0)
                hps._check(hps.LIB.hps_shard_session_last_timing(shards[r], C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), C.byref(recv), C.byref(kb)))
                info[r][it] = (att.value, cap.value, kb.value, recv.value, sum(sent), [x.value for x in t])
                ref = O.np_lookup(tables, q, [q.size], [2.0])
                if not np.array_equal(_bits(out.cpu().numpy()), _bits(ref)):
                    errs.append((r, it, "mismatch"))
        except Exception as e:  # noqa: BLE001
            errs.append((r, "exception", repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs
    for r in range(P):
        assert info[r][0][0] == 2, info[r][0]                       # skew: one repetition, never more
        assert all(info[r][it][0] <= 2 for it in range(4))
        assert info[r][1][0] == 1                                   # the capacity found is kept
        assert info[r][0][1] == info[0][0][1] and info[r][0][1] > max_local // P   # every rank chose the same capacity
        assert info[r][0][2] == 4 and info[r][2][2] == 8           # narrowed when the keys fit, 8 bytes with wide keys
        assert info[r][2][4] == max_local - int(np.count_nonzero(rounds[2][r] == np.iinfo(np.int64).min))   # the reserved key was not sent
        assert all(x >= 0 for x in info[r][3][5])
    assert sum(info[r][1][3] for r in range(P)) == P * max_local   # keys received over all ranks = keys sent
    for h in shards:
        hps.LIB.hps_shard_session_destroy(h)
    hps.LIB.hps_shard_group_destroy(grp)


@pytest.mark.gpu
def test_native_sharded_session_a_failing_rank_releases_the_others():
    """A rank that cannot take part in a collective call (here: more keys than max_local_keys) aborts its endpoint: the
    other rank's call returns an error instead of waiting for ever, and the group stays unusable."""
    import ctypes as C
    import threading
    import torch
    from hugectr_backend_amd import hps
    P = 2
    tables, made = _native_shards(P)
    keys = tables[0][0]
    grp = C.c_void_p()
    hps._check(hps.LIB.hps_shard_group_create_local(P, C.byref(grp)))
    shards = []
    for r, (_, sess) in enumerate(made):
        h = C.c_void_p()
        hps._check(hps.LIB.hps_shard_session_create_local(sess._h, grp, r, 1000, C.byref(h)))
        shards.append(h)
    rc = [None, None]

    def work(r):
        torch.cuda.set_device(0)
        n = 1000 if r == 0 else 1001
        dq = torch.from_numpy(np.resize(keys, n).astype(np.int64)).cuda()
        out = torch.empty(n * 128, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        rc[r] = hps.LIB.hps_shard_session_lookup(shards[r], dq.data_ptr(), n, out.data_ptr())

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=60) for t in th]
    assert not any(t.is_alive() for t in th), "a rank is still waiting for the one that failed"
    assert rc[1] == hps.ERR_INVALID_ARG and rc[0] not in (0, None)
    for h in shards:
        hps.LIB.hps_shard_session_destroy(h)
    hps.LIB.hps_shard_group_destroy(grp)
    for _, sess in made:
        sess.close()


@pytest.mark.gpu
def test_native_sharded_session_an_oversized_block_is_refused_on_every_rank_and_the_session_lives_on():
    """A request that puts more keys on one shard than a block may hold (block limit = lookup-session capacity / shards) is
    found out AFTER the key exchange, from the block headers every rank received: all ranks return the same error together
    and nobody is left waiting — so the transport is NOT aborted (round 3 did: one skewed request killed the sharded session
    on all ranks until every session was recreated).  The next request that fits is served exactly."""
    import ctypes as C
    import threading
    import torch
    from hugectr_backend_amd import hps, sharded
    from oracle import hps_oracle as O
    P = 2
    tables, made = _native_shards(P, max_local=1000)        # lookup sessions hold 2,000 keys: block limit 1,000
    keys = tables[0][0]
    owner = sharded.owner_of(keys, P)
    grp = C.c_void_p()
    hps._check(hps.LIB.hps_shard_group_create_local(P, C.byref(grp)))
    shards = []
    for r, (_, sess) in enumerate(made):
        h = C.c_void_p()
        hps._check(hps.LIB.hps_shard_session_create_local(sess._h, grp, r, 1500, C.byref(h)))
        shards.append(h)
    rng = np.random.default_rng(5)
    skewed = [rng.choice(keys[owner == 0], 1500).astype(np.int64) for _ in range(P)]     # 1,500 keys for shard 0 from each rank
    fair = [rng.choice(keys, 1500).astype(np.int64) for _ in range(P)]
    rc = [[None, None, None] for _ in range(P)]
    errs = []

    def work(r):
        torch.cuda.set_device(0)
        for it, q in enumerate((skewed[r], fair[r], skewed[r])):
            dq = torch.from_numpy(q).cuda()
            out = torch.empty(q.size * 128, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            rc[r][it] = hps.LIB.hps_shard_session_lookup(shards[r], dq.data_ptr(), q.size, out.data_ptr())
            if it == 1 and rc[r][it] == 0:
                ref = O.np_lookup(tables, q, [q.size], [2.0])
                if not np.array_equal(_bits(out.cpu().numpy()), _bits(ref)):
                    errs.append((r, "rows differ"))

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th)
    assert not errs, errs
    for r in range(P):
        assert rc[r][0] == hps.ERR_INVALID_ARG and rc[r][2] == hps.ERR_INVALID_ARG, rc
        assert rc[r][1] == 0, rc                 # the session survived the refusal
    for h in shards:
        hps.LIB.hps_shard_session_destroy(h)
    hps.LIB.hps_shard_group_destroy(grp)
    for _, sess in made:
        sess.close()



@pytest.mark.gpu
def test_native_sharded_session_zipf_request_ships_each_row_once_and_the_blocks_shrink():
    """Input dedup in front of the exchange (K1 before the bucket step): a Zipf request — a handful of keys make up most of it —
    sends every distinct key ONCE, so the hot key that overflowed a block in round 3 no longer does (one attempt), and because
    blocks travel whole, the block capacity follows the traffic down: after 32 calls that all fitted a smaller capacity the
    session ships blocks of the size the traffic needs.  Asserted: keys sent = distinct keys of the request, the row bytes per
    peer (capacity x D x 4) drop well below what the same request cost as sent, every rank resizes in the same call, rows exact
    throughout — and a later request of distinct keys that needs the old capacity again is served (one repeated call)."""
    import ctypes as C
    import threading
    import torch
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    P = 4
    max_local = 40000
    tables, made = _native_shards(P, R=60000, max_local=max_local)
    keys = tables[0][0]
    grp = C.c_void_p()
    hps._check(hps.LIB.hps_shard_group_create_local(P, C.byref(grp)))
    shards = []
    for r, (_, sess) in enumerate(made):
        h = C.c_void_p()
        hps._check(hps.LIB.hps_shard_session_create_local(sess._h, grp, r, max_local, C.byref(h)))
        shards.append(h)
    for _, sess in made:
        sess.close()
    rng = np.random.default_rng(91)
    ranks = np.argsort(rng.random(keys.size))
    w = 1.0 / np.arange(1, keys.size + 1) ** 1.3
    w /= w.sum()
    CALLS = 36
    rounds = [[keys[ranks[rng.choice(keys.size, size=max_local, p=w)]].astype(np.int64) for _ in range(P)] for _ in range(CALLS)]
    # last round: all-distinct keys again (what the first capacity was made for)
    rounds.append([rng.choice(keys, size=max_local, replace=False).astype(np.int64) for _ in range(P)])
    errs = []
    info = [[None] * len(rounds) for _ in range(P)]

    def work(r):
        try:
            torch.cuda.set_device(0)
            for it, per_rank in enumerate(rounds):
                q = per_rank[r]
                out = torch.empty(q.size * 128, dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                hps._check(hps.LIB.hps_shard_session_lookup_host(shards[r], q.ctypes.data, q.size, out.data_ptr()))
                att, cap = C.c_uint32(0), C.c_uint64(0)
                sent = (C.c_uint64 * P)()
                hps._check(hps.LIB.hps_shard_session_last_stats(shards[r], C.byref(cap), C.byref(att), sent, P))
                info[r][it] = (att.value, cap.value, sum(sent))
                if it % 6 == 0 or it >= CALLS - 1:
                    ref = O.np_lookup(tables, q, [q.size], [2.0])
                    if not np.array_equal(_bits(out.cpu().numpy()), _bits(ref)):
                        errs.append((r, it, "mismatch"))
        except Exception as e:  # noqa: BLE001
            errs.append((r, "exception", repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errs, errs
    cap0 = info[0][0][1]
    for r in range(P):
        for it in range(CALLS):
            assert info[r][it][0] == 1, (r, it, info[r][it])                                  # nothing overflows
            assert info[r][it][2] == np.unique(rounds[it][r]).size, (r, it)                  # each distinct key travels once
            assert info[r][it][1] == info[0][it][1], (r, it)                                  # same capacity on every rank, call by call
        assert info[r][31][1] == cap0 and info[r][32][1] < 0.7 * cap0, (info[r][31], info[r][32], cap0)   # resized after 32 calls
        largest = max(max(np.bincount(hps_owner(rounds[it][q_], P), minlength=P)) for it in range(32) for q_ in range(P))
        assert info[r][32][1] >= largest                                                       # ... to what the traffic needed
        # the all-distinct request no longer fits: ONE repeated call with the capacity it needs, rows exact (checked above)
        assert info[r][CALLS][0] == 2 and info[r][CALLS][1] > info[r][32][1]
    # bytes: what one rank ships to one peer per call (blocks travel whole)
    as_sent = cap0 * 128 * 4
    deduped = info[0][32][1] * 128 * 4
    assert deduped < 0.7 * as_sent
    for h in shards:
        hps.LIB.hps_shard_session_destroy(h)
    hps.LIB.hps_shard_group_destroy(grp)


def hps_owner(q, P):
    """owners of the DISTINCT keys of a request (what travels with the dedup on)"""
    from hugectr_backend_amd import sharded
    return sharded.owner_of(np.unique(q), P)
