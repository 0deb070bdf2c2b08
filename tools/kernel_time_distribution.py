import json, subprocess, sys, numpy as np
sys.path.insert(0, '.')
import bench as B
# monkeypatch summarize to dump distributions
orig = B.summarize
def summ(rec, N, D, dt, steps):
    a = np.array([[r[1], r[2], r[3], r[4]] for r in rec]) * 1e3
    for i, n in enumerate(["probe", "gather", "scatter", "insert"]):
        c = a[:, i]
        sys.stderr.write("KDIST %s mean %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f\n" % (n, c.mean(), *np.percentile(c, [10, 50, 90, 99]), c.max()))
    return orig(rec, N, D, dt, steps)
B.summarize = summ
sys.argv = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extra-legs", "--no-cpu-baseline", "--no-triton-leg"]
B.main()
