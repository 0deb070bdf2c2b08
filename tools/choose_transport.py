#!/usr/bin/env python3
"""First contact with a multi-GPU node: read ONE `python bench.py --gpus N` result and say which `"shard_transport"` to ship.

    python bench.py --gpus 8 --steps 20 --warmup 5          # writes bench_extra.json next to it
    python tools/choose_transport.py [bench_extra.json]      # the steps of INTEGRATION.md 4.1, applied to the numbers

No multi-GPU node was in reach of this project in any round, so the table-sharded model (BASELINE config 3 behind one Triton instance,
csrc/cache/shard_entry.h) ships with TWO transports of the rows and this script to pick one from the first run on real hardware:

  1. the self-test (`hps_multi_gpu_selftest`): a timeout or a failed step names what to fix on the platform first; without peer access
     only `staged_copy` is available;
  2. what ONE link gives a kernel's stores and a copy engine (`pair_GBps_median`);
  3. the two transports on the same requests: one request at a time and a request on every instance at once; how close the rows'
     rate into the entry GPU comes to (devices - 1) links of its mechanism; for `staged_copy`, whether its owners wait for copies
     (smaller pieces) or pay piece overheads (larger pieces);
  4. the SPMD variant's exchange rate against a link.

Output: a short report on stdout and, as its last line, one JSON object {"shard_transport": ..., "shard_copy_piece_keys": ..., "confidence":
..., "reasons": [...]}.  Exit status 0 when a choice could be made, 1 when the platform has to be fixed first or the run carries no
multi-GPU measurement (all shards on one device), 2 when the file cannot be read.  Pure Python over the JSON; tested in the CPU suite with
synthetic results (tests/test_choose_transport.py).
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

MARGIN = 0.03          # a transport has to be this much faster to be called faster
AUTO_PIECE = 131072    # csrc/cache/shard_entry.h: kAutoPieceKeys


def _get(d, *path, default=None):
    for p in path:
        if not isinstance(d, dict) or p not in d or d[p] is None:
            return default
        d = d[p]
    return d


def decide(full: dict) -> dict:
    """full = the object bench.py writes to bench_extra.json.  Returns the decision object (see module docstring) + "report" lines."""
    lines, reasons = [], []
    out = {"shard_transport": None, "shard_copy_piece_keys": 0, "confidence": "none", "reasons": reasons, "report": lines, "ok": False}
    st = full.get("multi_gpu_selftest")
    leg = _get(full, "extra_legs", "sharded_c3_single_entry")
    # ---- 1. the platform ----
    if not isinstance(st, dict):
        reasons.append("no multi_gpu_selftest in the result: run `python bench.py --gpus N` with N > 1 (without --no-selftest)")
        return out
    devs = st.get("devices") or []
    lines.append(f"self-test over devices {devs}: {st.get('seconds')} s")
    if st.get("timeout"):
        reasons.append(f"the self-test did not finish inside its deadline, stuck in '{st.get('stuck_in')}': fix the platform first "
                       "(IOMMU / ACS settings, HSA_ENABLE_IPC_MODE_LEGACY=0, the RCCL build) — nothing below it can be trusted")
        return out
    if st.get("error"):
        reasons.append(f"the self-test failed: {st.get('error')}")
        return out
    if len(set(devs)) < 2:
        reasons.append("every shard sat on ONE device (logical shards): the owners' stores and copies stayed in local HBM, no link was "
                       "measured — the numbers compare code paths, not transports")
        return out
    pa = st.get("peer_access") or []
    ok4 = st.get("store_4k_ok") or []
    n = len(devs)
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
    no_peer = [(devs[i], devs[j]) for i, j in pairs if not (i < len(pa) and j < len(pa[i]) and pa[i][j])]
    bad_store = [(devs[i], devs[j]) for i, j in pairs if (i < len(pa) and j < len(pa[i]) and pa[i][j]) and not (i < len(ok4) and j < len(ok4[i]) and ok4[i][j])]
    store_ok = not no_peer and not bad_store
    if no_peer:
        lines.append(f"no peer access for {len(no_peer)} ordered pair(s), e.g. {no_peer[:3]}: peer_store is NOT available")
    if bad_store:
        lines.append(f"a 4-KB peer store did not arrive for {bad_store[:3]}: peer_store must not be used on this platform")
    rc = st.get("rccl_allreduce") or {}
    lines.append(f"RCCL all-reduce over {rc.get('ranks')} rank(s): {'ok' if rc.get('ok') else 'FAILED: ' + str(rc.get('error'))} "
                 f"({rc.get('ms')} ms with communicator set-up)")
    # ---- 2. one link ----
    link_store = _get(st, "pair_GBps_median", "store")
    link_copy = _get(st, "pair_GBps_median", "copy")
    lines.append(f"one link, median over the ordered pairs: kernel stores {link_store} GB/s, copy engine {link_copy} GB/s "
                 f"(slowest pair: {_get(st, 'pair_GBps_min', 'store')} / {_get(st, 'pair_GBps_min', 'copy')})")
    # ---- 3. the two transports on the same requests ----
    if not isinstance(leg, dict) or not isinstance(leg.get("by_transport"), dict):
        reasons.append("the result has no sharded_c3_single_entry.by_transport leg (run without --no-extra-legs)")
        return out
    bt = leg["by_transport"]
    rate = {}
    for name in ("peer_store", "staged_copy"):
        u = _get(bt, name, "uniform")
        if not isinstance(u, dict) or "error" in u:
            lines.append(f"{name}: not measured ({_get(bt, name, 'error') or _get(u or {}, 'error')})")
            continue
        if not u.get("parity"):
            lines.append(f"{name}: ROWS WRONG (parity false) — not a candidate")
            continue
        one = u.get("lookups_per_s") or 0.0
        allv = _get(u, "all_instances_at_once", "lookups_per_s") or 0.0
        into = u.get("rows_GBps_into_entry_gpu")
        link = link_store if name == "peer_store" else link_copy
        frac = (into / ((n - 1) * link)) if (into and link) else None
        rate[name] = (one, allv)
        lines.append(f"{name}: {one / 1e9:.3f} G lookups/s one request at a time, {allv / 1e9:.3f} G with a request on every instance; rows into "
                     f"the entry GPU {into} GB/s" + (f" = {frac:.2f} of {n - 1} links of its mechanism" if frac is not None else ""))
    if not store_ok:
        rate.pop("peer_store", None)
    if not rate:
        reasons.append("neither transport produced a valid measurement")
        return out
    if len(rate) == 1:
        pick = next(iter(rate))
        reasons.append(f"only {pick} is available / valid on this node")
        conf = "forced"
    else:
        # the serving case is every instance busy; one request at a time breaks a tie
        s_one, s_all = rate["peer_store"]
        c_one, c_all = rate["staged_copy"]
        key_s, key_c = (s_all or s_one), (c_all or c_one)
        if key_c > key_s * (1 + MARGIN):
            pick, conf = "staged_copy", "measured"
            reasons.append(f"staged_copy is {key_c / key_s:.2f} x peer_store with a request on every instance")
        elif key_s > key_c * (1 + MARGIN):
            pick, conf = "peer_store", "measured"
            reasons.append(f"peer_store is {key_s / key_c:.2f} x staged_copy with a request on every instance")
        else:
            pick = "peer_store" if s_one >= c_one else "staged_copy"
            conf = "tie"
            reasons.append(f"within {MARGIN:.0%} of each other with every instance busy; {pick} is ahead one request at a time "
                           f"({s_one / 1e9:.3f} against {c_one / 1e9:.3f} G)")
    out.update(shard_transport=pick, confidence=conf, ok=True)
    # ---- staged_copy: the piece size ----
    u = _get(bt, "staged_copy", "uniform")
    if pick == "staged_copy" and isinstance(u, dict):
        wait, slow = u.get("copy_wait_ms_slowest_shard"), u.get("slowest_shard_ms")
        pieces = max(u.get("pieces_per_shard") or [1])
        if wait is not None and slow:
            share = wait / slow
            lines.append(f"staged_copy: the slowest owner spends {share:.0%} of its {slow:.3f} ms waiting for copies, {pieces} piece(s) per shard")
            if share > 0.5 and pieces <= 4:
                out["shard_copy_piece_keys"] = AUTO_PIECE // 2
                reasons.append("its owners wait for copies more than half of their time: smaller pieces (more overlap) — try 65,536")
            elif share < 0.2 and pieces > 8:
                out["shard_copy_piece_keys"] = AUTO_PIECE * 2
                reasons.append("its owners hardly wait for copies and cut the bucket into many pieces: larger pieces (fewer piece overheads) — try 262,144")
            else:
                reasons.append("piece size: keep the automatic one")
    # ---- 4. the SPMD variant ----
    sp = _get(full, "extra_legs", "sharded_c3")
    if isinstance(sp, dict) and "error" not in sp:
        gbps = sp.get("row_exchange_GBps_per_rank")
        lines.append(f"SPMD variant ({sp.get('ranks')} ranks, {sp.get('backend')}): {((sp.get('lookups_per_s') or 0) / 1e9):.3f} G lookups/s, rows exchanged at "
                     f"{gbps} GB/s per rank over the whole step" + (f" (one link's copy rate: {link_copy} GB/s)" if link_copy else ""))
    return out


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    path = Path(argv[0]) if argv else Path("bench_extra.json")
    try:
        full = json.loads(path.read_text())
    except (OSError, ValueError) as e:
        print(f"cannot read {path}: {e}")
        return 2
    d = decide(full)
    for l in d.pop("report"):
        print(l)
    for r in d["reasons"]:
        print("=> " + r)
    ok = d.pop("ok")
    print(json.dumps(d))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
