#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries committed under profiles/.

    python tools/summarize_profile.py <dir with kt/ pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <out.json>
    python tools/summarize_profile.py --latest <rocprof_summary.json> profiles/pmc_latest.json <round> "<source>"

Kernel stats: top rows of *_kernel_stats.csv (names shortened).  PMC: mean FETCH_SIZE / WRITE_SIZE of every kernel
of a lookup call, converted as MI355X_MICROARCH.md §HBM prescribes: both counters are in KiB; on gfx950
FETCH_SIZE counts 128-B coalesced requests as 64 B, so wide streaming reads are doubled.  Reported both ways.
"""
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")[:80]


def split_insert(d):
    out = {}
    for f in glob.glob(f"{d}/kt/**/*kernel_trace.csv", recursive=True) + glob.glob(f"{d}/kt/*kernel_trace.csv"):
        rows = list(csv.DictReader(open(f)))
        if not rows:
            continue
        probes = [int(r["Start_Timestamp"]) for r in rows if "hps_probe_tile" in r["Kernel_Name"]]
        first_probe = min(probes) if probes else None
        groups = {"warm_up_before_the_first_request": [], "serving": []}
        for r in rows:
            if "hps_cache_insert" not in r["Kernel_Name"]:
                continue
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
            key = "warm_up_before_the_first_request" if first_probe is None or int(r["Start_Timestamp"]) < first_probe else "serving"
            groups[key].append((dur, grid))
        for k, v in groups.items():
            if v:
                ds = sorted(x[0] for x in v)
                out[k] = {"launches": len(v), "avg_us": sum(ds) / len(ds), "median_us": ds[len(ds) // 2], "min_us": ds[0], "max_us": ds[-1],
                          "grid_sizes": sorted({x[1] for x in v})}
        break
    return out


def main(d, out):
    res = {"kernel_stats": [], "pmc": {}}
    for f in glob.glob(f"{d}/kt/*kernel_stats.csv"):
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            if "hps::" in r["Name"]:
                res["kernel_stats"].append({"kernel": short(r["Name"]), "calls": int(r["Calls"]),
                                            "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                            "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])})
    # The insert kernel's launches fall into two populations that the statistics table averages together: the cache WARM-UP at
    # model load (262,144-row chunks of EmbeddingCache::InsertKeys, before any request) and the SERVING calls (72 K missed rows
    # each).  Split by time: everything before the first probe launch is warm-up (rocprofv3's own average of round 5 — 87.9 us —
    # was dominated by the former; the timed steps' inserts take ~34 us).
    res["insert_kernel_split"] = split_insert(d)

    def pmc_of(kernel_pat):
        got = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            vals = []
            for f in glob.glob(f"{d}/pmc_{c}/*counter_collection.csv"):
                rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
                # serving launches only: what was dispatched before the first probe is the cache warm-up (the insert kernel's
                # 262,144-row chunks at model load — 559 MB per launch in round 5's table against ~80 MB for a call's 72 K rows)
                first = min((int(r["Dispatch_Id"]) for r in rows if "hps_probe_tile" in r["Kernel_Name"] and r.get("Dispatch_Id")), default=None)
                for r in rows:
                    if kernel_pat in r["Kernel_Name"] and (first is None or not r.get("Dispatch_Id") or int(r["Dispatch_Id"]) >= first):
                        vals.append(float(r["Counter_Value"]))
            if vals:
                got[c] = {"launches": len(vals), "mean_KiB": sum(vals) / len(vals), "min_KiB": min(vals), "max_KiB": max(vals)}
        if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
            f, w = got["FETCH_SIZE"]["mean_KiB"] * 1024, got["WRITE_SIZE"]["mean_KiB"] * 1024
            got["hbm_bytes_per_launch_raw"] = f + w
            got["hbm_bytes_per_launch_fetch_doubled"] = 2 * f + w
        return got

    kernels = ("hps_probe_tile", "hps_miss_unique", "hps_gather_hits", "hps_miss_scatter", "hps_cache_insert")
    res["pmc_by_kernel"] = {k: v for k, v in ((k, pmc_of(k)) for k in kernels) if v}
    # SURVEY 8(d)'s lookup = probe + unique + gather + scatter (the insert is cache maintenance, reported on its own)
    # (hps_miss_unique is a launch of its own only when the unique-hit count is needed: since round 3 its work is the tail of
    #  hps_probe_tile and its bytes are counted there)
    call = [res["pmc_by_kernel"].get(k, {}) for k in kernels[:4]]
    if all("hbm_bytes_per_launch_fetch_doubled" in res["pmc_by_kernel"].get(k, {}) for k in ("hps_probe_tile", "hps_gather_hits")):
        res["pmc"] = {
            "hbm_bytes_per_call_fetch_doubled": sum(c.get("hbm_bytes_per_launch_fetch_doubled", 0.0) for c in call),
            "hbm_bytes_per_call_raw": sum(c.get("hbm_bytes_per_launch_raw", 0.0) for c in call),
            "note": "sum over hps_probe_tile (+ hps_miss_unique where it is a launch of its own) + hps_gather_hits + hps_miss_scatter, one launch of each per "
                    "lookup call, one session; FETCH_SIZE and WRITE_SIZE from separate --pmc passes, both in KiB; on gfx950 "
                    "FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled (MI355X_MICROARCH.md, HBM) — the bulk of "
                    "the reads are 16 B/lane row segments and 128-B bucket lines"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:1500])


def write_latest(summary_json, latest_json, round_no, source):
    """profiles/pmc_latest.json: what bench.py copies into roofline.traffic (it cannot run the profiler on itself)."""
    r = json.load(open(summary_json))
    json.dump({"round": int(round_no), "workload_keys": 1703936, "dim": 128, "source": source, "pmc": r["pmc"],
               "pmc_by_kernel": r["pmc_by_kernel"], "insert_kernel_split": r.get("insert_kernel_split")}, open(latest_json, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "--latest":   # --latest <rocprof_summary.json> <profiles/pmc_latest.json> <round> <source text>
        write_latest(*sys.argv[2:6])
    else:
        main(sys.argv[1], sys.argv[2])
