"""Condense a rocprofv3 kernel trace (…_kernel_trace.csv) into what the scheduling questions need:
per-kernel count/avg/min/max over the steady-state tail, the busy fraction of a chosen kernel (union of its
intervals over the window), and a short textual timeline.

usage: python tools/timeline.py <kernel_trace.csv> [busy-kernel-substring] [tail_fraction]
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    busy_pat = sys.argv[2] if len(sys.argv) > 2 else "ps_fetch_direct"
    tail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48],
                         r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    only = [r for r in rows if "hps" in r[2]]
    if only:  # restrict to this library's kernels and to the span they cover
        rows = only
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t1 - int((t1 - t0) * tail)
    marks = sorted(r[0] for r in rows if busy_pat in r[2])
    hi = t1
    if len(marks) >= 16:  # steady state: from the first quarter of the marker kernel's launches to its third-last
        lo, hi = marks[len(marks) // 4], marks[-3]
    win = [r for r in rows if lo <= r[0] < hi]
    agg = defaultdict(list)
    for s, e, n, _ in win:
        agg[n].append(e - s)
    span = (max(r[1] for r in win) - win[0][0]) / 1e3
    print(f"window {span/1e3:.2f} ms, {len(win)} kernels")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {n:48s} n={len(v):5d} avg={sum(v)/len(v)/1e3:8.1f} us min={min(v)/1e3:8.1f} max={max(v)/1e3:8.1f} "
              f"sum={sum(v)/1e3/span*100:5.1f}% of window")
    iv = sorted((s, e) for s, e, n, _ in win if busy_pat in n)
    busy, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            busy += ce - cs
            cs, ce = s, e
    if cs is not None:
        busy += ce - cs
    print(f"busy fraction of '{busy_pat}': {busy/1e3/span:.3f}  (sum of its durations / union = "
          f"{sum(e - s for s, e in iv)/max(busy, 1):.2f} = average overlap depth)")
    base = win[len(win) // 2][0]
    print("timeline (us from an arbitrary steady-state origin):")
    for s, e, n, q in win[len(win) // 2: len(win) // 2 + 60]:
        print(f"  {(s-base)/1e3:9.1f} -> {(e-base)/1e3:9.1f}  ({(e-s)/1e3:7.1f})  q{q}  {n}")


if __name__ == "__main__":
    main()
