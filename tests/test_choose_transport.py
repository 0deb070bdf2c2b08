"""tools/choose_transport.py: the steps of INTEGRATION.md 4.1 applied to a `bench.py --gpus N` result.  No multi-GPU node was ever in
reach, so the tool is tested on synthetic results shaped like bench_extra.json (keys as bench.py writes them: multi_gpu_selftest,
extra_legs.sharded_c3_single_entry.by_transport.<transport>.uniform, extra_legs.sharded_c3)."""
import copy
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import choose_transport as ct  # noqa: E402


def _leg(rate, all_rate, into, wait=0.0, slow=1.0, pieces=1, parity=True):
    return {"uniform": {"parity": parity, "lookups_per_s": rate, "rows_GBps_into_entry_gpu": into, "copy_wait_ms_slowest_shard": wait,
                        "slowest_shard_ms": slow, "pieces_per_shard": [pieces] * 4,
                        "all_instances_at_once": {"instances": 4, "lookups_per_s": all_rate, "parity": True}}}


def _result(n=4, store=(2.0e9, 6.0e9, 300.0), copy_=(1.5e9, 5.0e9, 250.0), **copy_kw):
    m = [[None if i == j else True for j in range(n)] for i in range(n)]
    return {"n_gpus": n,
            "multi_gpu_selftest": {"devices": list(range(n)), "peer_access": m, "store_4k_ok": copy.deepcopy(m), "timeout": False, "stuck_in": None,
                                   "error": None, "seconds": 3.0, "pair_GBps_median": {"store": 45.0, "copy": 48.0},
                                   "pair_GBps_min": {"store": 40.0, "copy": 47.0}, "rccl_allreduce": {"ranks": n, "ok": True, "ms": 900.0, "error": None}},
            "extra_legs": {"sharded_c3_single_entry": {"by_transport": {"peer_store": _leg(*store), "staged_copy": _leg(*copy_, **copy_kw)}},
                           "sharded_c3": {"ranks": n, "backend": "nccl", "lookups_per_s": 3.0e9, "row_exchange_GBps_per_rank": 120.0}}}


def test_the_faster_transport_with_every_instance_busy_is_chosen():
    d = ct.decide(_result())
    assert d["ok"] and d["shard_transport"] == "peer_store" and d["confidence"] == "measured"
    assert any("0.6" in l or "2.2" in l for l in d["report"] if "links of its mechanism" in l)   # 300 / (3 x 45) = 2.22 (a synthetic number)
    d = ct.decide(_result(copy_=(1.5e9, 7.0e9, 330.0)))
    assert d["shard_transport"] == "staged_copy" and d["confidence"] == "measured" and "1.17 x" in d["reasons"][0]


def test_a_tie_goes_to_the_one_ahead_on_single_requests():
    d = ct.decide(_result(store=(2.0e9, 6.0e9, 300.0), copy_=(2.2e9, 6.1e9, 300.0)))
    assert d["shard_transport"] == "staged_copy" and d["confidence"] == "tie"


def test_piece_size_hint_follows_the_owners_wait_for_copies():
    d = ct.decide(_result(copy_=(2.5e9, 8.0e9, 330.0), wait=0.7, slow=1.0, pieces=2))
    assert d["shard_transport"] == "staged_copy" and d["shard_copy_piece_keys"] == 65536
    d = ct.decide(_result(copy_=(2.5e9, 8.0e9, 330.0), wait=0.05, slow=1.0, pieces=13))
    assert d["shard_copy_piece_keys"] == 262144
    d = ct.decide(_result(copy_=(2.5e9, 8.0e9, 330.0), wait=0.3, slow=1.0, pieces=4))
    assert d["shard_copy_piece_keys"] == 0 and any("automatic" in r for r in d["reasons"])


def test_no_peer_access_or_a_lost_store_forces_staged_copy():
    r = _result()
    r["multi_gpu_selftest"]["peer_access"][2][1] = False
    d = ct.decide(r)
    assert d["ok"] and d["shard_transport"] == "staged_copy" and d["confidence"] == "forced"
    r = _result()
    r["multi_gpu_selftest"]["store_4k_ok"][0][3] = False
    d = ct.decide(r)
    assert d["shard_transport"] == "staged_copy" and any("did not arrive" in l for l in d["report"])


def test_wrong_rows_disqualify_a_transport():
    r = _result(copy_=(9e9, 9e9, 900.0))
    r["extra_legs"]["sharded_c3_single_entry"]["by_transport"]["staged_copy"]["uniform"]["parity"] = False
    d = ct.decide(r)
    assert d["shard_transport"] == "peer_store" and d["confidence"] == "forced"


def test_a_stuck_selftest_or_logical_shards_give_no_choice(tmp_path):
    r = _result()
    r["multi_gpu_selftest"].update(timeout=True, stuck_in="peer store 0 -> 3")
    d = ct.decide(r)
    assert not d["ok"] and d["shard_transport"] is None and "peer store 0 -> 3" in d["reasons"][0]
    r = _result()
    r["multi_gpu_selftest"]["devices"] = [0]
    d = ct.decide(r)
    assert not d["ok"] and "ONE device" in d["reasons"][0]
    # the command line: last line = the decision object, exit status 1 without a choice, 0 with one, 2 without a file
    p = tmp_path / "bench_extra.json"
    p.write_text(json.dumps(r))
    run = subprocess.run([sys.executable, str(ROOT / "tools" / "choose_transport.py"), str(p)], capture_output=True, text=True)
    assert run.returncode == 1 and json.loads(run.stdout.strip().splitlines()[-1])["shard_transport"] is None
    p.write_text(json.dumps(_result()))
    run = subprocess.run([sys.executable, str(ROOT / "tools" / "choose_transport.py"), str(p)], capture_output=True, text=True)
    assert run.returncode == 0 and json.loads(run.stdout.strip().splitlines()[-1])["shard_transport"] == "peer_store"
    assert subprocess.run([sys.executable, str(ROOT / "tools" / "choose_transport.py"), str(tmp_path / "none.json")], capture_output=True).returncode == 2


def test_the_one_gpu_box_result_of_this_round_parses():
    """The shape bench.py really writes (a trimmed copy of a `--gpus 2` run on the one-GPU box): no choice, no exception."""
    real = {"multi_gpu_selftest": {"devices": [0], "probe_bytes": 67108864, "peer_access": [[None]], "store_4k_ok": [[None]], "store_GBps": [[None]],
                                   "copy_GBps": [[None]], "pair_GBps_min": {"store": None, "copy": None}, "pair_GBps_median": {"store": None, "copy": None},
                                   "rccl_allreduce": {"ranks": 1, "ok": True, "ms": 3414.81, "error": None}, "error": None, "timeout": False,
                                   "stuck_in": None, "seconds": 4.04},
            "extra_legs": {"sharded_c3_single_entry": {"by_transport": {
                "peer_store": {"uniform": {"parity": True, "lookups_per_s": 2.02e9, "rows_GBps_into_entry_gpu": None, "pieces_per_shard": [1, 1],
                                           "copy_wait_ms_slowest_shard": 0.0, "slowest_shard_ms": 0.546,
                                           "all_instances_at_once": {"instances": 2, "lookups_per_s": 3.24e9, "parity": True}}},
                "staged_copy": {"uniform": {"parity": True, "lookups_per_s": 1.12e9, "rows_GBps_into_entry_gpu": None, "pieces_per_shard": [1, 1],
                                            "copy_wait_ms_slowest_shard": 0.367, "slowest_shard_ms": 1.219}}}}}}
    d = ct.decide(real)
    assert not d["ok"] and d["shard_transport"] is None
