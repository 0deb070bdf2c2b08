#!/usr/bin/env python3
"""Merged GPU timeline (kernels + memory copies) of a rocprofv3 run, from the middle of the library's activity.

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o t -- python bench.py ...
    python tools/gpu_timeline.py DIR [events=120] [position=0.6]

Prints, for `events` consecutive events starting at `position` of the span of this library's kernels: start, end, duration,
queue / stream id, idle gap since the previous event ended (device-wide), and the name (copies with size and GB/s); then the
device-idle share of that window."""
import csv
import glob
import sys


def load(d):
    ev = []
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                       r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hps::", "")[:44],
                       "q" + str(r.get("Queue_Id", "?")) + "/s" + str(r.get("Stream_Id", "?")), 0))
    for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "")),
                       "s" + str(r.get("Stream_Id", "?")), int(r.get("Size", r.get("Bytes", 0)) or 0)))
    return sorted(ev)


def main():
    ev = load(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    pos = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
    probes = [i for i, e in enumerate(ev) if "probe_tile" in e[2]]
    if not probes:
        print("no probe kernels in the trace")
        return
    i0 = probes[int(len(probes) * pos)]
    win = ev[i0:i0 + n]
    base = win[0][0]
    end_prev = win[0][0]
    idle = 0
    for s, e, name, q, size in win:
        gap = s - end_prev
        if gap > 0:
            idle += gap
        extra = f"  {size/1e6:7.2f} MB {size/max(e-s,1):6.1f} GB/s" if size else ""
        print(f"{(s-base)/1e3:9.1f} -> {(e-base)/1e3:9.1f} ({(e-s)/1e3:7.1f} us)  {q:10s} gap {max(gap,0)/1e3:6.1f}  {name}{extra}")
        end_prev = max(end_prev, e)
    span = end_prev - base
    print(f"window {span/1e3:.1f} us, device idle {idle/1e3:.1f} us = {idle/max(span,1):.2%}")


if __name__ == "__main__":
    main()
