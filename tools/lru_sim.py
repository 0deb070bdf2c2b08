#!/usr/bin/env python3
"""CPU model of ONE table's GPU cache policy over thousands of calls (the GPU bench sees 260): 14-slot buckets, one-byte recency
stamps in units of 2^age_shift calls, insert age, admission rule (csrc/cache/kernels.hip, hps_cache_insert_kernel) — to see where
the hit rate settles once every bucket has been full for a long time.

    python tools/lru_sim.py --calls 4000 --admit 0,4,15 [--insert-age 32] [--age-shift 3]

Workload = bench.py's: per call B keys, with probability `hit` a Zipf(1.05)-ranked key of the C rows resident after warm-up, else
a cold key that never returns.  Simplifications: ages are not taken modulo 255 (no wrap, no saturation), one insert per bucket
and call wins ties by order; the "current unit is never evicted" and "newly inserted is never evicted in the same call" rules
are kept.
"""
import argparse
import time

import numpy as np


def zipf_cdf(n, s):
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), s)
    c = np.cumsum(w)
    return c / c[-1]


def run(a, admit_log2):
    rng = np.random.default_rng(7)
    C = int(a.rows * a.cache_frac)
    slots = int(np.ceil(C / 0.75))
    nb = (slots + 13) // 14
    keys = np.full((nb, 14), -1, np.int64)
    stamp = np.zeros((nb, 14), np.int64)          # call unit of the last hit (or nominal unit of insertion)
    mult = np.uint64(0x9E3779B97F4A7C15)

    def bucket_of(k):
        h = (k.astype(np.uint64) + np.uint64(1)) * mult
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
        return ((h >> np.uint64(32)) * np.uint64(nb) >> np.uint64(32)).astype(np.int64)

    # warm-up: the first C keys in order, dropped where the bucket is full
    wk = np.arange(C, dtype=np.int64)
    wb = bucket_of(wk)
    order = np.argsort(wb, kind="stable")
    wb, wk = wb[order], wk[order]
    start = np.searchsorted(wb, np.arange(nb))
    pos = np.arange(C) - start[wb]
    ok = pos < 14
    keys[wb[ok], pos[ok]] = wk[ok]
    resident = np.sort(keys[keys >= 0])
    cdf = zipf_cdf(resident.size, a.zipf)
    next_cold = a.rows     # cold keys: a counter, never repeated
    hits_hist = []
    t0 = time.time()
    for call in range(1, a.calls + 1):
        now = call >> a.age_shift
        ins_unit = now - a.insert_age
        hot = rng.random(a.batch) < a.hit
        nh = int(hot.sum())
        q_hot = resident[np.minimum(np.searchsorted(cdf, rng.random(nh)), resident.size - 1)]
        ncold = a.batch - nh
        q_cold = np.arange(next_cold, next_cold + ncold, dtype=np.int64)
        next_cold += ncold
        # hot keys: hit if still resident
        b = bucket_of(q_hot)
        eq = keys[b] == q_hot[:, None]
        is_hit = eq.any(axis=1)
        slot = eq.argmax(axis=1)
        stamp[b[is_hit], slot[is_hit]] = now
        hits_hist.append(int(is_hit.sum()) / a.batch)
        # misses: unique evicted hot keys + all cold keys
        miss = np.concatenate([np.unique(q_hot[~is_hit]), q_cold])
        mb = bucket_of(miss)
        bypass = ((miss * 2654435761 >> 8) ^ call) & ((1 << admit_log2) - 1) == 0 if admit_log2 > 0 else np.ones(miss.size, bool)
        inserted_here = {}
        for k, bb, bp in zip(miss.tolist(), mb.tolist(), bypass.tolist()):
            row_k, row_s = keys[bb], stamp[bb]
            age = np.where(row_k < 0, 1 << 40, now - row_s)
            age[row_s == now] = -1                                   # current unit: never evicted
            done = inserted_here.get(bb)
            if done:
                age[done] = -1                                       # inserted by this call: off limits
            if admit_log2 > 0 and a.insert_age > 0 and not bp:
                age[(age >= 0) & (age < a.insert_age)] = -1          # admission: younger than the newcomer's nominal age
            v = int(age.argmax())
            if age[v] < 0:
                continue                                             # dropped
            row_k[v] = k
            row_s[v] = ins_unit
            inserted_here.setdefault(bb, []).append(v)
        if call % a.report == 0:
            print(f"  admit={admit_log2:2d} call {call:5d}: hit rate of the last {a.report} calls {np.mean(hits_hist[-a.report:]):.4f}  "
                  f"({time.time() - t0:.0f} s)", flush=True)
    return hits_hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=3000)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cache-frac", type=float, default=0.2)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--hit", type=float, default=0.957)
    ap.add_argument("--zipf", type=float, default=1.05)
    ap.add_argument("--age-shift", type=int, default=3)
    ap.add_argument("--insert-age", type=int, default=32)
    ap.add_argument("--admit", default="0,4")
    ap.add_argument("--report", type=int, default=250)
    a = ap.parse_args()
    for adm in [int(x) for x in a.admit.split(",")]:
        h = run(a, adm)
        print(f"admit={adm}: first 260 calls {np.mean(h[:260]):.4f}, last {a.report} calls {np.mean(h[-a.report:]):.4f}")


if __name__ == "__main__":
    main()
