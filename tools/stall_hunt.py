#!/usr/bin/env python3
"""Where does a multi-millisecond host stall of bench.py's timed region sit on the GPU's timeline?

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o t -- python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline
    python tools/stall_hunt.py <dir>

Merges the kernel and memory-copy traces, restricts them to the span of this library's kernels, and prints (1) the
longest memory copies with their rate, (2) the longest intervals in which NOTHING ran on the device, with what ended before
and what started after each.  A stalled DMA shows up under (1), a stalled host thread under (2).
"""
import csv
import glob
import sys


def load(d):
    ev = []
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:50], 0))
    for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy")), int(r.get("Size", r.get("Bytes", 0)) or 0)))
    return sorted(ev)


def main():
    ev = load(sys.argv[1])
    ours = [e for e in ev if "hps::" in e[2]]
    probes = sorted(e[0] for e in ours if "probe_tile" in e[2])
    lo, hi = probes[len(probes) // 8], probes[-2]          # skip table warm-up and the tail
    win = [e for e in ev if lo <= e[0] <= hi]
    print(f"window {(hi - lo) / 1e6:.1f} ms, {len(win)} events, {len(probes)} probe launches")
    copies = sorted((e for e in win if e[2].startswith("C ")), key=lambda e: e[0] - e[1])[:8]
    print("longest copies:")
    for s, e, n, b in copies:
        print(f"  {(s - lo) / 1e6:9.3f} ms  {(e - s) / 1e3:9.1f} us  {b / 1e6:8.2f} MB  {b / max(e - s, 1):6.1f} GB/s  {n}")
    # idle gaps: sweep over the union of all intervals
    gaps, cur_end, last = [], win[0][1], win[0]
    for ev_ in win[1:]:
        if ev_[0] > cur_end:
            gaps.append((ev_[0] - cur_end, cur_end, last, ev_))
        if ev_[1] > cur_end:
            cur_end, last = ev_[1], ev_
    gaps.sort(key=lambda g: -g[0])
    print("longest device-idle gaps:")
    for g, at, before, after in gaps[:8]:
        print(f"  {(at - lo) / 1e6:9.3f} ms  idle {g / 1e3:9.1f} us   after [{before[2]}]   before [{after[2]}]")
    busy = sum(min(e[1], hi) - e[0] for e in win) / (hi - lo)
    print(f"sum of event durations / window = {busy:.2f}")


if __name__ == "__main__":
    main()
