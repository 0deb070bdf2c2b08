#!/usr/bin/env python3
"""How busy is the host->device link in the timed region, and what runs on the GPU while it idles?

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o t -- python bench.py ... --no-extra-legs --no-cpu-baseline
    python tools/link_busy.py DIR [tail_fraction=0.5]

Over the last `tail_fraction` of the library's activity: union of the H2D copy intervals / window (= link busy), bytes / window
(= rate over the window) and bytes / busy time (= rate while copying); the 12 longest link-idle gaps with the kernels that ran inside
each; and the distribution of copy sizes and rates."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    ker, cop = [], []
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hps::", "")
            if "hps_" in n:
                ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:40]))
    for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "HOST_TO_DEVICE" in r.get("Direction", r.get("Name", "")):
                size = 0
                for col in ("Size", "Bytes", "Size_Bytes", "Copy_Size", "size"):
                    if r.get(col):
                        size = int(r[col])
                        break
                cop.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), size))
    ker.sort(), cop.sort()
    # the timed region's steady state: between the probe launches at (1 - tail) and at 0.95 of all probe launches
    probes = [k[0] for k in ker if "probe_tile" in k[2]]
    lo, t1 = probes[int(len(probes) * (1.0 - tail))], probes[int(len(probes) * 0.95)]
    cw = [c for c in cop if c[0] >= lo and c[1] <= t1]
    busy, gaps, cs, ce = 0, [], None, None
    for s, e, _ in cw:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            busy += ce - cs
            gaps.append((s - ce, ce, s))
            cs, ce = s, e
    if cs is not None:
        busy += ce - cs
    span = cw[-1][1] - cw[0][0]
    nbytes = sum(c[2] for c in cw)
    print(f"window {span/1e6:.2f} ms, {len(cw)} H2D copies, {nbytes/1e6:.1f} MB")
    print(f"link busy {busy/span:.3f} of the window; {nbytes/span:.2f} GB/s over the window; {nbytes/max(busy,1):.2f} GB/s while copying")
    big = [c for c in cw if c[2] >= (1 << 20)]
    if big:
        rates = sorted(c[2] / max(c[1] - c[0], 1) for c in big)
        print(f"copies >= 1 MB: {len(big)}, rate p10 {rates[len(rates)//10]:.1f} p50 {rates[len(rates)//2]:.1f} p90 {rates[len(rates)*9//10]:.1f} GB/s "
              f"(sum of their durations / union of all copies = {sum(c[1]-c[0] for c in cw)/max(busy,1):.2f} = overlap depth)")
    tot_gap = sum(g[0] for g in gaps)
    print(f"idle: {tot_gap/span:.3f} of the window in {len(gaps)} gaps; gaps > 20 us: {sum(g[0] for g in gaps if g[0] > 20000)/span:.3f}")
    for g, a, b in sorted(gaps, reverse=True)[:12]:
        inside = [k for k in ker if k[1] > a and k[0] < b]
        names = ", ".join(f"{k[2].replace('_kernel','').replace('hps_','')}({(min(k[1],b)-max(k[0],a))/1e3:.0f})" for k in inside[:8])
        print(f"  gap {g/1e3:7.1f} us at +{(a-cw[0][0])/1e6:8.3f} ms: {names}")


if __name__ == "__main__":
    main()
