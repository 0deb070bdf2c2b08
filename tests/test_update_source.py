"""Online update source (ps.json "update_source"): the consumer loop of the reference's real-time update path
(/root/reference/hps_backend/src/backend.cpp:262-308, docs/hierarchical_parameter_server.md:575-646) over this build's
file-tail transport — messages appended to a file are dispatched to the host tier in chunks, committed, survive a restart
exactly once; Kafka stays refused."""
import json
import os

import numpy as np
import pytest

from tests.conftest import make_tables, ps_config


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _server(tmp_path, tables, src, gpucache=False, **kw):
    from hugectr_backend_amd import hps
    cfg = ps_config("upd", tables, gpucache=gpucache, **kw)
    cfg["update_source"] = src
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays("upd", t, k, r)
    return ps


def test_kafka_is_refused_and_file_tail_needs_a_path():
    from hugectr_backend_amd import hps
    tables = make_tables([(100, 4)])
    for src, needle in (({"type": "kafka_message_queue", "brokers": "127.0.0.1:9092"}, "Kafka"),
                        ({"type": "file_tail", "brokers": ""}, "path")):
        cfg = ps_config("upd", tables, gpucache=False)
        cfg["update_source"] = src
        with pytest.raises(hps.HpsError, match=needle):
            hps.HierParameterServer.create_from_dict(cfg, load_tables=False)


def test_messages_reach_the_host_tier_in_chunks_and_are_committed_once(tmp_path):
    from hugectr_backend_amd import hps
    rng = np.random.default_rng(5)
    tables = make_tables([(3000, 8), (500, 16)])
    path = tmp_path / "updates.bin"
    src = {"type": "file_tail", "brokers": str(path), "poll_timeout_ms": 50, "max_batch_size": 64, "max_commit_interval": 3,
           "failure_backoff_ms": 5}
    ps = _server(tmp_path, tables, src)
    sess = hps.LookupSession.create(ps, "upd", None)
    # expected content, updated as the messages are written
    want = [dict(zip(k.tolist(), map(np.array, r))) for k, r in tables]
    with open(path, "ab") as f:
        for i in range(7):
            t = i % 2
            k = np.concatenate([rng.choice(tables[t][0], 150), 10_000_000 + rng.integers(0, 1000, 20)]).astype(np.int64)   # overwrite + new keys
            k = np.unique(k)
            r = rng.random((k.size, tables[t][1].shape[1]), dtype=np.float32)
            f.write(hps.encode_update_message("upd", t, k, r))
            f.flush()
            for kk, rr in zip(k.tolist(), r):
                want[t][kk] = rr
        # a message for a model this server does not hold, and one with the wrong row width: dropped, not fatal
        f.write(hps.encode_update_message("someone_else", 0, np.arange(3), np.zeros((3, 8), np.float32)))
        f.write(hps.encode_update_message("upd", 0, np.arange(3), np.zeros((3, 5), np.float32)))
        # half a frame: the producer is still writing
        tail = hps.encode_update_message("upd", 1, np.array([42], np.int64), np.full((1, 16), 7.5, np.float32))
        f.write(tail[:20])
        f.flush()
        ps.drain_update_source(10000)
        st = ps.update_source_stats()
        assert st["messages"] == 7 and st["rejected_messages"] == 2 and st["commits"] >= 3
        assert st["dispatches"] >= 7 * 3      # ~170 keys per message in chunks of 64
        for t in range(2):
            q = np.array(list(want[t].keys()), dtype=np.int64)
            nk = [q.size, 0] if t == 0 else [0, q.size]
            out = sess.lookup(q, nk).reshape(q.size, -1)
            assert np.array_equal(_bits(out), _bits(np.stack([want[t][int(x)] for x in q])))
        # the rest of the frame arrives
        f.write(tail[20:])
        f.flush()
        ps.drain_update_source(10000)
    out = sess.lookup(np.array([42], np.int64), [0, 1])
    assert np.all(out == 7.5)
    assert ps.update_source_stats()["messages"] == 8
    committed = int(open(str(path) + ".offset").read())
    assert committed == os.path.getsize(path)
    sess.close()
    ps.close()
    # a restarted server resumes behind the last commit: nothing is applied twice, new messages are
    ps2 = _server(tmp_path, tables, src)
    with open(path, "ab") as f:
        f.write(hps.encode_update_message("upd", 0, np.array([7], np.int64), np.full((1, 8), -2.0, np.float32)))
    ps2.drain_update_source(10000)
    st = ps2.update_source_stats()
    assert st["messages"] == 1 and st["keys"] == 1
    s2 = hps.LookupSession.create(ps2, "upd", None)
    assert np.all(s2.lookup(np.array([7], np.int64), [1, 0]) == -2.0)
    s2.close()
    ps2.close()


def test_a_frame_that_cannot_be_a_message_stops_the_source_without_taking_the_server_down(tmp_path):
    from hugectr_backend_amd import hps
    tables = make_tables([(200, 4)])
    path = tmp_path / "updates.bin"
    path.write_bytes(b"this is not a message frame at all, just text....")
    ps = _server(tmp_path, tables, {"type": "file_tail", "brokers": str(path), "poll_timeout_ms": 20, "failure_backoff_ms": 5})
    # a dead source is not "drained": the call says what is wrong with it
    with pytest.raises(hps.HpsError, match="not a message"):
        ps.drain_update_source(3000)
    import time
    time.sleep(0.3)   # dozens of polls later the frame is still counted ONCE
    assert ps.update_source_stats()["rejected_messages"] == 1 and ps.update_source_stats()["messages"] == 0
    sess = hps.LookupSession.create(ps, "upd", None)
    q = tables[0][0][:10]
    assert np.array_equal(_bits(sess.lookup(q, [10]).reshape(10, 4)), _bits(tables[0][1][:10]))
    sess.close()


def test_a_well_formed_message_over_the_receive_buffer_is_skipped_and_the_source_lives_on(tmp_path):
    """receive_buffer_size bounds one message; a producer's max_batch_size of 8,192 keys can easily exceed a small buffer.
    Such a frame has a valid header and a known length: it is stepped over (one rejection), later messages are applied."""
    from hugectr_backend_amd import hps
    tables = make_tables([(300, 8)])
    path = tmp_path / "updates.bin"
    src = {"type": "file_tail", "brokers": str(path), "poll_timeout_ms": 20, "failure_backoff_ms": 5, "receive_buffer_size": 2048}
    ps = _server(tmp_path, tables, src)
    k = tables[0][0]
    with open(path, "ab") as f:
        f.write(hps.encode_update_message("upd", 0, k[:10], np.full((10, 8), 1.5, np.float32)))          # 10 * (8 + 32) = 400 B: fits
        f.write(hps.encode_update_message("upd", 0, k[10:260], np.full((250, 8), 2.5, np.float32)))      # 10,000 B: too large (the bound is never below 4 KB)
        f.write(hps.encode_update_message("upd", 0, k[260:270], np.full((10, 8), 3.5, np.float32)))      # fits again
    ps.drain_update_source(10000)
    st = ps.update_source_stats()
    assert st["messages"] == 2 and st["rejected_messages"] == 1 and st["keys"] == 20
    sess = hps.LookupSession.create(ps, "upd", None)
    out = sess.lookup(k[:270], [270]).reshape(270, 8)
    assert np.all(out[:10] == 1.5) and np.all(out[260:] == 3.5)
    assert np.array_equal(_bits(out[10:260]), _bits(tables[0][1][10:260]))     # the skipped message changed nothing
    assert int(open(str(path) + ".offset").read()) == os.path.getsize(path)
    # ... and it is still alive
    with open(path, "ab") as f:
        f.write(hps.encode_update_message("upd", 0, k[:1], np.full((1, 8), 4.5, np.float32)))
    ps.drain_update_source(10000)
    assert np.all(sess.lookup(k[:1], [1]) == 4.5)
    sess.close()


def test_a_consumer_stopped_in_the_middle_of_a_backlog_commits_only_what_it_applied(tmp_path):
    """Advisor finding of round 3: with the stop flag set, the messages of a poll that had not been applied yet were counted
    and the offset moved past them — after a restart those updates were gone.  Now: stop = no further count, no commit.
    Every message in front of the committed offset IS in the host tier of the server that committed it, and a restarted
    server applies every message behind it."""
    import threading
    import time
    from hugectr_backend_amd import hps
    tables = make_tables([(2000, 8)])
    path = tmp_path / "updates.bin"
    src = {"type": "file_tail", "brokers": str(path), "poll_timeout_ms": 20, "max_batch_size": 8, "max_commit_interval": 32,
           "failure_backoff_ms": 5}
    ends, blob = [], bytearray()
    NMSG, PER = 3000, 64
    for i in range(NMSG):      # every message brings keys of its own: a message that is lost leaves a hole nothing fills
        k = 10_000_000 + i * PER + np.arange(PER, dtype=np.int64)
        blob += hps.encode_update_message("upd", 0, k, np.full((PER, 8), float(i + 1), np.float32))
        ends.append(len(blob))
    q = 10_000_000 + np.arange(NMSG * PER, dtype=np.int64)

    def applied(ps_):
        s_ = hps.LookupSession.create(ps_, "upd", None)
        out = s_.lookup(q, [q.size]).reshape(NMSG, PER, 8)
        s_.close()
        return np.array([bool(np.all(out[i] == float(i + 1))) for i in range(NMSG)])

    ps = _server(tmp_path, tables, src)      # (tables loaded: from here on messages for "upd" are applied)

    def produce():
        with open(path, "ab") as f:
            f.write(blob)

    w = threading.Thread(target=produce)
    w.start()
    deadline = time.time() + 30
    while ps.update_source_stats()["messages"] < 200 and time.time() < deadline:   # somewhere inside the backlog
        time.sleep(0.0002)
    ps.stop_update_source()
    w.join()
    off_file = str(path) + ".offset"
    committed = int(open(off_file).read()) if os.path.exists(off_file) else 0
    assert committed == 0 or committed in ends
    n_committed = ends.index(committed) + 1 if committed else 0
    assert n_committed < NMSG, "the backlog was meant to outlast the first consumer (make it longer)"
    got = applied(ps)
    assert got[:n_committed].all(), f"committed past messages that never reached the host tier: {np.flatnonzero(~got[:n_committed])[:10]}"
    with pytest.raises(hps.HpsError):
        ps.update_source_stats()          # no consumer any more
    ps.close()
    # the next server starts behind the commit and applies every message behind it (its host tier starts from the model files:
    # what the first server held in RAM is gone with it, which is why the offset must never run ahead)
    ps2 = _server(tmp_path, tables, src)
    ps2.drain_update_source(60000)
    assert int(open(off_file).read()) == ends[-1]
    assert ps2.update_source_stats()["messages"] == NMSG - n_committed
    assert applied(ps2)[n_committed:].all()
    ps2.close()


@pytest.mark.gpu
def test_resident_rows_are_replaced_in_the_gpu_cache_and_absent_keys_stay_out(tmp_path):
    from hugectr_backend_amd import hps
    tables = make_tables([(4000, 16)])
    path = tmp_path / "updates.bin"
    src = {"type": "file_tail", "brokers": str(path), "poll_timeout_ms": 30, "max_batch_size": 256}
    ps = _server(tmp_path, tables, src, gpucache=True, gpucacheper=0.25, hit_rate_threshold=1.0)
    ps.create_embedding_cache_per_model("upd")
    cache = ps.get_embedding_cache("upd", 0)
    sess = hps.LookupSession.create(ps, "upd", cache)
    keys = tables[0][0]
    res = keys[cache.query(0, keys) >= 0]
    cold = keys[cache.query(0, keys) < 0]
    assert res.size > 100 and cold.size > 100
    upd = np.concatenate([res[:64], cold[:64]])
    with open(path, "ab") as f:
        f.write(hps.encode_update_message("upd", 0, upd, np.full((upd.size, 16), 9.0, np.float32)))
    ps.drain_update_source(10000)
    # the cold keys were NOT pulled into the cache by the update; the resident ones answer with the new rows from the cache
    assert np.all(cache.query(0, cold[:64]) < 0) and np.all(cache.query(0, res[:64]) >= 0)
    before = cache.counters()["misses"]
    out = sess.lookup(res[:64], [64]).cpu().numpy()
    assert np.all(out == 9.0) and cache.counters()["misses"] == before
    out = sess.lookup(cold[:64], [64]).cpu().numpy()     # ... and the host tier has the new rows for the others
    assert np.all(out == 9.0)
    sess.close()


@pytest.mark.gpu
def test_stopping_the_consumer_delivers_what_it_applied_to_the_gpu_caches(tmp_path):
    """Advisor finding of round 4: hps_server_update_source_stop returned without passing the messages applied since the last
    commit to the GPU caches — the server keeps serving after the stop, and resident keys answered with their OLD rows while
    the host tier already held the new ones.  A commit interval and a poll timeout the test never reaches keep everything
    'applied, not committed' until the stop."""
    import time
    from hugectr_backend_amd import hps
    tables = make_tables([(4000, 16)])
    path = tmp_path / "updates.bin"
    src = {"type": "file_tail", "brokers": str(path), "poll_timeout_ms": 60000, "max_commit_interval": 100000, "max_batch_size": 256}
    ps = _server(tmp_path, tables, src, gpucache=True, gpucacheper=0.25, hit_rate_threshold=1.0)
    ps.create_embedding_cache_per_model("upd")
    cache = ps.get_embedding_cache("upd", 0)
    sess = hps.LookupSession.create(ps, "upd", cache)
    keys = tables[0][0]
    res = keys[cache.query(0, keys) >= 0][:64]
    with open(path, "ab") as f:
        f.write(hps.encode_update_message("upd", 0, res, np.full((res.size, 16), 11.0, np.float32)))
    t0 = time.time()
    while ps.update_source_stats()["messages"] < 1 and time.time() - t0 < 20:
        time.sleep(0.01)
    st = ps.update_source_stats()
    assert st["messages"] == 1 and st["commits"] == 0          # applied to the host tier, not committed, caches not told yet
    ps.stop_update_source()
    out = sess.lookup(res, [res.size]).cpu().numpy()
    assert np.all(out == 11.0), "resident keys still answer with the rows from before the update"
    assert np.all(cache.query(0, res) >= 0)
    # ... and the offset moved past exactly that message
    assert int(open(str(path) + ".offset").read()) == path.stat().st_size
    sess.close()


def test_update_filters_decide_which_updates_the_database_layers_take(tmp_path):
    """volatile_db.update_filters / persistent_db.update_filters (docs/hierarchical_parameter_server.md:509-512): regular
    expressions over the update's tag hps_<model>.<table name>.  Round 3 parsed and ignored them."""
    from hugectr_backend_amd import hps
    tables = make_tables([(100, 4), (100, 4)])
    path = tmp_path / "updates.bin"
    src = {"type": "file_tail", "brokers": str(path), "poll_timeout_ms": 20, "failure_backoff_ms": 5}
    cfg = ps_config("upd", tables, gpucache=False)
    cfg["update_source"] = src
    cfg["models"][0]["embedding_table_names"] = ["user", "item"]
    cfg["volatile_db"]["update_filters"] = ["^hps_upd\\.item$", "^hps_somebody_else\\..+$"]
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays("upd", t, k, r)
    with open(path, "ab") as f:
        f.write(hps.encode_update_message("upd", 0, tables[0][0][:5], np.full((5, 4), 3.0, np.float32)))   # hps_upd.user: filtered out
        f.write(hps.encode_update_message("upd", 1, tables[1][0][:5], np.full((5, 4), 4.0, np.float32)))   # hps_upd.item: taken
    ps.drain_update_source(10000)
    st = ps.update_source_stats()
    # a message nobody subscribed to is skipped silently (round 4 counted it as a failed dispatch and a rejected message)
    assert st["messages"] == 1 and st["rejected_messages"] == 0 and st["dispatch_failures"] == 0
    assert ps.filtered_update_count() == 1
    sess = hps.LookupSession.create(ps, "upd", None)
    out = sess.lookup(np.concatenate([tables[0][0][:5], tables[1][0][:5]]), [5, 5]).reshape(10, 4)
    assert np.array_equal(_bits(out[:5]), _bits(tables[0][1][:5])) and np.all(out[5:] == 4.0)
    sess.close()
    ps.close()
    # not a regular expression: refused at start-up, not ignored
    cfg["volatile_db"]["update_filters"] = ["("]
    with pytest.raises(hps.HpsError, match="regular expression"):
        hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    # lists that differ between the layers are fine (round 4 refused them, even [".+"] against the default "^hps_.+$" which
    # select the same updates): every layer takes the updates its own list selects (the next test); a table that is ONE store —
    # here: the whole table in memory, unbounded volatile database — is updated when either list selects it
    cfg["volatile_db"]["update_filters"] = ["^hps_upd\\.item$"]
    cfg["persistent_db"] = {"type": "rocks_db", "path": str(tmp_path / "store"), "update_filters": ["^hps_upd\\.user$", "("]}
    with pytest.raises(hps.HpsError, match="persistent_db.update_filters"):
        hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    cfg["persistent_db"]["update_filters"] = ["^hps_upd\\.user$"]
    for t, (k, r) in enumerate(tables):
        O_dir = tmp_path / f"files_{t}"
        from oracle import hps_oracle as O
        O.np_write_table(O_dir, k, r)
        cfg["models"][0]["sparse_files"][t] = str(O_dir)
    path2 = tmp_path / "updates2.bin"
    cfg["update_source"] = dict(src, brokers=str(path2))
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=True)
    with open(path2, "ab") as f:
        f.write(hps.encode_update_message("upd", 0, tables[0][0][:5], np.full((5, 4), 3.0, np.float32)))   # persistent list takes it
        f.write(hps.encode_update_message("upd", 1, tables[1][0][:5], np.full((5, 4), 4.0, np.float32)))   # volatile list takes it
    ps.drain_update_source(10000)
    assert ps.update_source_stats()["messages"] == 2 and ps.filtered_update_count() == 0
    ps.close()


def test_each_database_layer_subscribes_with_its_own_update_filters(tmp_path):
    """Advisor finding of round 5: differing volatile_db / persistent_db update_filters were merged — an update only the persistent
    database subscribed to also changed the volatile tier (and the GPU caches).  In the reference each layer subscribes with its
    own filter (backend.cpp:207-216, 250-259).  Bounded volatile tier in front of a writable row store, every key cached at load
    time: a persistent-only update lands in the row store and the volatile tier keeps answering with the row it holds; a
    volatile-only update is served at once and never reaches the row store."""
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    tables = make_tables([(60, 4), (60, 4)])
    path = tmp_path / "updates.bin"
    cfg = ps_config("upd", tables, gpucache=False)
    cfg["update_source"] = {"type": "file_tail", "brokers": str(path), "poll_timeout_ms": 20, "failure_backoff_ms": 5}
    cfg["models"][0]["embedding_table_names"] = ["user", "item"]
    cfg["volatile_db"].update({"overflow_margin": 1000, "overflow_policy": "evict_oldest", "initial_cache_rate": 1.0,
                               "update_filters": ["^hps_upd\\.item$"]})
    cfg["persistent_db"] = {"type": "rocks_db", "path": str(tmp_path / "store"), "update_filters": ["^hps_upd\\.user$"]}
    for t, (k, r) in enumerate(tables):
        O.np_write_table(tmp_path / f"files_{t}", k, r)
        cfg["models"][0]["sparse_files"][t] = str(tmp_path / f"files_{t}")
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=True)
    assert ps.host_tier_stats("upd", 0)["tiered"] == 1
    ku, ki = tables[0][0][:5], tables[1][0][:5]
    with open(path, "ab") as f:
        f.write(hps.encode_update_message("upd", 0, ku, np.full((5, 4), 3.0, np.float32)))   # hps_upd.user: persistent layer only
        f.write(hps.encode_update_message("upd", 1, ki, np.full((5, 4), 4.0, np.float32)))   # hps_upd.item: volatile layer only
    ps.drain_update_source(10000)
    assert ps.update_source_stats()["messages"] == 2 and ps.filtered_update_count() == 0
    sess = hps.LookupSession.create(ps, "upd", None)
    out = sess.lookup(np.concatenate([ku, ki]), [5, 5]).reshape(10, 4)
    assert np.array_equal(_bits(out[:5]), _bits(tables[0][1][:5])), "the volatile tier took an update it did not subscribe to"
    assert np.all(out[5:] == 4.0)
    sess.close()
    ps.close()
    # what is on disk afterwards: the row store of `user` has the new rows, the one of `item` the old ones
    rows = {}
    for ev in (tmp_path / "store").rglob("emb_vector"):
        k = np.fromfile(ev.parent / "key", np.int64)
        r = np.fromfile(ev, np.float32).reshape(k.size, 4)
        rows[str(ev.parent)] = dict(zip(k.tolist(), r))
    user = [m for m in rows.values() if int(ku[0]) in m and np.all(m[int(ku[0])] == 3.0)]
    item_old = [m for m in rows.values() if int(ki[0]) in m and np.array_equal(m[int(ki[0])], tables[1][1][0])]
    assert user, "the persistent layer did not take the update it subscribed to"
    assert item_old, "the persistent layer took an update only the volatile database subscribed to"
