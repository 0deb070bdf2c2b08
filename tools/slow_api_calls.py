#!/usr/bin/env python3
"""Which runtime call is the slow one?  Scans rocprofv3 API traces (HIP runtime + HSA core/AMD-extension, csv) for calls
longer than a threshold and prints them in time order with the calls of the other threads that were in flight meanwhile.

    rocprofv3 --hip-runtime-trace --hsa-core-trace --hsa-amd-trace --output-format csv -d DIR -- python bench.py ...
    python tools/slow_api_calls.py DIR [--ms 2.0]
"""
import argparse
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--ms", type=float, default=2.0)
    ap.add_argument("--skip", default="hipStreamSynchronize,hipEventSynchronize,hipDeviceSynchronize,hsa_signal_wait_scacquire,hsa_signal_wait_relaxed,hipMemcpy,hipHostMalloc,hipMalloc,hipFree,hipHostFree,hipModuleLoad",
                    help="functions that are expected to block (comma list, exact names)")
    a = ap.parse_args()
    skip = set(a.skip.split(","))
    rows = []
    for f in glob.glob(os.path.join(a.dir, "**", "*_api_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                try:
                    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                except (KeyError, ValueError):
                    continue
                rows.append((s, e, r.get("Function", "?"), r.get("Thread_Id", "?"), r.get("Domain", "?")))
    rows.sort()
    if not rows:
        print("no api trace rows under", a.dir)
        return
    t0 = rows[0][0]
    slow = [r for r in rows if (r[1] - r[0]) / 1e6 >= a.ms and r[2] not in skip]
    print(f"{len(rows)} calls, {len(slow)} longer than {a.ms} ms (expected blockers skipped)")
    by_name = {}
    for s, e, fn, th, dom in slow:
        by_name.setdefault(fn, []).append((e - s) / 1e6)
    for fn, v in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {fn:48s} n={len(v):4d} total {sum(v):9.1f} ms max {max(v):7.2f} ms")
    setup = {"hsa_amd_memory_pool_allocate", "hsa_amd_memory_pool_free", "hipGetDeviceCount", "hipStreamCreateWithFlags", "hsa_queue_create",
             "hsa_amd_agents_allow_access", "hipStreamDestroy", "hsa_executable_freeze", "hsa_executable_load_agent_code_object"}
    last_setup = max([e for s_, e, fn, th, dom in rows if fn in ("hipStreamCreateWithFlags", "hipHostMalloc", "hipMalloc")] or [t0])
    print(f"last stream creation / allocation ends at t={(last_setup - t0) / 1e6:.1f} ms; slow calls that are not set-up calls (time since first call):")
    for s, e, fn, th, dom in slow:
        if fn in setup:
            continue
        print(f"  t={(s - t0) / 1e6:10.1f} ms  {(e - s) / 1e6:7.2f} ms  thread {th}  {dom}:{fn}")
        # what else was in flight (other threads) inside this window
        n = 0
        for s2, e2, fn2, th2, dom2 in rows:
            if s2 > e:
                break
            if th2 != th and e2 > s and (e2 - s2) / 1e6 >= 1.0:
                print(f"        meanwhile thread {th2}: {dom2}:{fn2} {(e2 - s2) / 1e6:.2f} ms (from t={(s2 - t0) / 1e6:.1f})")
                n += 1
                if n >= 6:
                    break


if __name__ == "__main__":
    main()
