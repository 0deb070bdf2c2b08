"""PCIe busy fraction of the host-gather tier from a rocprofv3 run with --kernel-trace --memory-copy-trace:
union of the host-to-device copies' intervals over the steady-state window of the bench's timed region, the same
for the probe+gather kernel, and a textual timeline of a few batches.

usage: python tools/copy_busy.py <dir with *_memory_copy_trace.csv and *_kernel_trace.csv>
"""
import csv
import glob
import sys


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    d = sys.argv[1]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    mt = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0]
    ka = []
    for r in csv.DictReader(open(kt)):
        if "hps_probe_gather" in r["Kernel_Name"]:
            ka.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    ka.sort()
    lo, hi = ka[len(ka) // 3][0], ka[-3][0]          # steady state of the timed region
    copies = []
    for r in csv.DictReader(open(mt)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        direction = r.get("Direction", r.get("Name", ""))
        copies.append((s, e, direction, int(r.get("Size", r.get("Bytes", 0)) or 0)))
    h2d = [(s, e) for s, e, dr, sz in copies if "HOST_TO_DEVICE" in dr.upper() and lo <= s < hi]
    h2d_bytes = sum(sz for s, e, dr, sz in copies if "HOST_TO_DEVICE" in dr.upper() and lo <= s < hi)
    win = hi - lo
    nb = sum(1 for s, e in ka if lo <= s < hi)
    busy = union(h2d)
    print(f"window {win / 1e6:.2f} ms, {nb} probe+gather launches ({win / 1e6 / max(nb, 1):.3f} ms per batch)")
    print(f"H2D copies: {len(h2d)} in window, {h2d_bytes / 1e6:.1f} MB ({h2d_bytes / 1e6 / max(nb, 1):.1f} MB per batch), "
          f"busy {busy / 1e6:.2f} ms = {busy / win:.3f} of the window, {h2d_bytes / max(busy, 1):.1f} GB/s while busy, "
          f"{h2d_bytes / win:.1f} GB/s over the window")
    print(f"probe+gather busy {union([(s, e) for s, e in ka if lo <= s < hi]) / win:.3f} of the window")
    # gaps between consecutive H2D copies larger than 30 us: where the link idles
    hs = sorted(h2d)
    gaps = [(b[0] - a[1]) for a, b in zip(hs, hs[1:]) if b[0] - a[1] > 30000]
    print(f"idle gaps > 30 us between copies: {len(gaps)}, total {sum(gaps) / 1e6:.2f} ms, "
          f"largest {max(gaps) / 1e3 if gaps else 0:.0f} us, median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.0f} us")


if __name__ == "__main__":
    main()
