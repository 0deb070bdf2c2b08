#!/usr/bin/env python3
"""Headline benchmark: embedding lookups/s on Criteo-shaped key batches (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the lookup hot path over one batch of synthetic keys that already sit in HBM
(26 tables x 65,536 keys): cache probe (HIP), unique-miss extraction (HIP), hit-row gather (HIP) running
while the parameter server fetches the missed rows (one GPU, >= 12 CPUs: host threads gather them and
hipMemcpyAsync ships them, as in the reference; otherwise / --direct 1: the device-driven tier, a HIP kernel
reading them out of pinned host memory, with the fused probe+gather kernel), scatter + cache insert (HIP).
The other tier and the fused kernel are measured after the timed region and reported under extra_legs.
Results are the exact fp32 rows (sync-insert mode, hit_rate_threshold=1.0), checked against the CPU
oracle on a slice of every run.

N>1 (launched by torch.distributed.run, one rank per GPU): the reference's multi-GPU mode is
"replicas only" (independent cache per GPU) — each rank serves its own batches, no
data-path collective; RCCL is used for the barriers and the max-over-ranks time only.  scaling = weak.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SEED = 20260929
PCIE_PEAK_GBS = 63.0   # MI355X_MICROARCH.md: PCIe Gen5 x16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tables", type=int, default=26)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per table (BASELINE config 2: 1e7)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536, help="samples per batch; one key per table per sample")
    ap.add_argument("--cache-frac", type=float, default=0.2, help="gpucacheper")
    ap.add_argument("--hit", type=float, default=0.957,
                    help="probability that a key is drawn from the set resident after warm-up (0.957 gives a MEASURED hit "
                         "rate of 0.95: inserting each batch's cold keys evicts a little of the hot tail)")
    ap.add_argument("--zipf", type=float, default=1.05)
    ap.add_argument("--sessions", type=int, default=2, help="concurrent lookup sessions (Triton instance count)")
    ap.add_argument("--mode", choices=["sync", "async"], default="sync",
                    help="sync: exact rows (threshold 1.0); async: misses return default, inserted in background")
    ap.add_argument("--distinct-batches", type=int, default=0,
                    help="0: one fresh batch per step (warmup+steps distinct batches)")
    ap.add_argument("--unroll", type=int, default=4, help="probe+gather kernel variant (tools/kbench.py)")
    ap.add_argument("--direct", type=int, default=-1,
                    help="parameter-server tier of the miss path.  0: host threads gather the missed rows and "
                         "hipMemcpyAsync ships them (the reference's arrangement); 1: ps_direct_access (the GPU resolves "
                         "misses itself out of pinned host memory, no host threads); -1 (default): host gather when this "
                         "rank has at least 12 CPUs to itself, else device-driven — on one GPU the other tier is measured "
                         "right after the headline and reported under extra_legs")
    ap.add_argument("--split-probe", type=int, default=-1,
                    help="1 / -1 (default): K_A probes only and the hit rows are moved by hps_gather_hits_kernel while the misses "
                         "are fetched (DESIGN.md 3.4c); 0: fused probe+gather kernel")
    ap.add_argument("--no-direct-leg", action="store_true",
                    help="one GPU, host-gather headline: skip the device-driven-tier leg measured afterwards")
    ap.add_argument("--no-sharded-leg", action="store_true",
                    help="N>1: skip the BASELINE config 3 leg (one table sharded over the ranks, RCCL all-to-all)")
    ap.add_argument("--shard-rows", type=int, default=1 << 28,
                    help="rows of the sharded table in total (config 3 names 1e9 = 512 GB; default 2^28 = 137 GB)")
    ap.add_argument("--setup-seconds", type=float, default=150.0,
                    help="N>1: bound on the estimated host-table generation time; rows/table shrinks to meet it")
    ap.add_argument("--sharded-steps", type=int, default=50)
    ap.add_argument("--sharded-timeout", type=float, default=300.0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the untimed all-hit and async-insert legs reported next to the headline")
    return ap.parse_args()


def effective_cpus() -> int:
    """CPUs this process may really use: affinity mask clipped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(np.ceil(int(q) / int(p)))))
    except Exception:
        pass
    return n


def cpu_throttle_stat():
    """(nr_throttled, throttled_usec) of this container's cgroup, or None: a run whose timed region was throttled by the
    CPU quota shows multi-millisecond stalls in its p99 that have nothing to do with the GPU path."""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except Exception:
        return None


def host_memory_budget() -> int:
    """Bytes of host RAM this container may still take: MemAvailable clipped by the cgroup limit."""
    avail = 1 << 62
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        if mx != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
            avail = min(avail, int(mx) - cur)
    except Exception:
        pass
    return avail


def zipf_cdf(n: int, alpha: float) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), alpha)
    c = np.cumsum(w)
    return c / c[-1]


def make_batches_gpu(torch, gen, resident_d, cdf_d, R, C, B, hit, nbatches):
    """Key batches generated on the device (every step gets fresh cold keys, so the hit rate is not
    inflated by re-running a batch whose misses were inserted the first time).
    per table: B keys; with prob `hit` a Zipf-ranked resident key, else uniform from the cold range [C,R)."""
    out = []
    for _ in range(nbatches):
        parts = []
        for res in resident_d:
            u = torch.rand(B, generator=gen, device="cuda", dtype=torch.float64)
            ranks = torch.searchsorted(cdf_d, u).clamp_(max=res.numel() - 1)
            hot = torch.rand(B, generator=gen, device="cuda") < hit
            if R > C:
                cold = torch.randint(C, R, (B,), generator=gen, device="cuda", dtype=torch.int64)
            else:
                cold = res[ranks]
            parts.append(torch.where(hot, res[ranks], cold))
        out.append(torch.cat(parts).contiguous())
    return out


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # share the host cores between the ranks' parameter-server pools
        os.environ.setdefault("HCTR_DEFAULT_CONCURRENCY", str(max(2, effective_cpus() // world)))
    tier_auto = a.direct < 0
    if tier_auto:
        # the host-gather tier needs host cores (its gather runs on ~14 threads per GPU); with fewer, or with several
        # replicas sharing one host, the device-driven tier, which needs none, is the one to run
        a.direct = 0 if (world == 1 and effective_cpus() >= 12) else 1

    # HIP spreads a process's streams over 4 hardware queues by default; two lookup sessions whose streams land on the
    # same queue run strictly one after the other (measured: 1.30 instead of 1.85 G lookups/s for the device-driven tier
    # after an earlier phase of the process had created other streams).  Must be in the environment before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    from hugectr_backend_amd.gpu_wait import wait_for_gpu
    wait_for_gpu(30.0)   # a device that another process has just released can be invisible for a moment
    import torch
    import torch.distributed as dist
    from hugectr_backend_amd import build as hb
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs an MI355X"
    # one rank per GPU over RCCL ("nccl" on ROCm).  Only when fewer GPUs than ranks are visible (the 1-GPU
    # development box) do the ranks share devices and rendezvous over gloo, to exercise the same control flow.
    shared_gpu = world > ndev
    dev = local_rank % ndev
    local_rank = dev
    torch.cuda.set_device(dev)
    coll_dev = "cpu" if shared_gpu else "cuda"
    if world > 1:
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    if rank == 0:
        hb.build()
    if world > 1:
        dist.barrier()  # the other ranks load the libraries only after rank 0 has (re)built them
    from hugectr_backend_amd import hps

    T, R, D, B = a.tables, a.rows, a.dim, a.batch
    # host-memory guard: every rank keeps the full tables in its parameter server (replicas).  If the box
    # cannot hold them for all ranks, rows/table shrinks and the workload string says so.
    rows_requested = R
    per_row = T * (4 * D + 64) + (2 * (4 * D + 48) if rank == 0 else 0)   # tables + index (+ oracle sample on rank 0)
    budget = int(host_memory_budget() * 0.85 / world) - (8 << 30)
    if world > 1:
        bt = torch.tensor([budget], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(bt, op=dist.ReduceOp.MIN)
        budget = int(bt.item())
    per_row = T * (4 * D + 64) + 2 * (4 * D + 48)
    if R * per_row > budget:
        R = max(B, int(budget // per_row))
    setup_note = ""
    if world > 1:
        # setup-time guard: N ranks generate and pin N copies of the tables on the host cores they share.  A probe
        # table measures this rank's generation rate under that contention; if the full tables would take longer
        # than --setup-seconds, rows/table shrinks for every rank (same batch, same hit rate; the workload says so).
        probe_rows = min(R, 1_000_000)
        pcfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
                "models": [{"model": "probe", "sparse_files": ["synthetic://probe"], "num_of_worker_buffer_in_pool": 1,
                            "embedding_vecsize_per_table": [D], "maxnum_catfeature_query_per_table_per_sample": [1],
                            "default_value_for_each_table": [0.0], "deployed_device_list": [local_rank],
                            "max_batch_size": 1, "gpucache": False}]}
        pps = hps.HierParameterServer.create_from_dict(pcfg, load_tables=False)
        dist.barrier()
        tp = time.time()
        pps.load_table_synthetic("probe", 0, SEED, 0, probe_rows)
        est = (time.time() - tp) / probe_rows * R * T * (1.6 if a.direct else 1.0)   # + page-locking
        del pps
        et = torch.tensor([est], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        est = float(et.item())
        if est > a.setup_seconds:
            R = max(B, int(R * a.setup_seconds / est) // 1000 * 1000)
            setup_note = f"; setup-time guard ({est:.0f} s estimated for the full tables on this box's shared host cores)"
    N = T * B
    model = "criteo_dlrm"
    cfg = {
        "supportlonglong": True,
        "volatile_db": {"type": "hash_map", "num_partitions": 8},
        "models": [{
            "model": model,
            "sparse_files": [f"synthetic://{t}" for t in range(T)],
            "num_of_worker_buffer_in_pool": max(3, a.sessions),
            "embedding_vecsize_per_table": [D] * T,
            "maxnum_catfeature_query_per_table_per_sample": [1] * T,
            "default_value_for_each_table": [0.0] * T,
            "deployed_device_list": [local_rank],
            "max_batch_size": B,
            "gpucache": True,
            "gpucacheper": a.cache_frac,
            "hit_rate_threshold": 1.0 if a.mode == "sync" else 0.5,
            "ps_direct_access": bool(a.direct),
        }],
    }
    def setup():
        t_setup = time.time()
        ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
        for t in range(T):
            ps.load_table_synthetic(model, t, SEED, 0, R)
        t_tables = time.time() - t_setup
        ps.create_embedding_cache_per_model(model)
        cache = ps.get_embedding_cache(model, local_rank)
        t_cache = time.time() - t_setup - t_tables
        return ps, cache, t_tables, t_cache

    # ps_direct_access pins the whole host tier (hipHostMalloc).  Where the box refuses that much page-locked
    # memory, every rank falls back to the host-gather tier of the same library (never to a CPU path) and the
    # result line says so in config.ps_tier.
    direct_note = None
    made = None
    if a.direct:
        try:
            made = setup()
        except Exception as e:  # noqa: BLE001
            direct_note = f"ps_direct_access unavailable on this box ({str(e)[:160]})"
            sys.stderr.write(f"[bench rank {rank}] {direct_note}; falling back to the host-gather tier\n")
        ok = 1 if made is not None else 0
        if world > 1:
            okt = torch.tensor([ok], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = int(okt.item())
        if not ok:
            made = None
            import gc
            gc.collect()
            a.direct = 0
            cfg["models"][0]["ps_direct_access"] = False
            direct_note = direct_note or "ps_direct_access unavailable on another rank"
    if made is None:
        made = setup()
    ps, cache, t_tables, t_cache = made
    sessions = [hps.LookupSession.create(ps, model, cache) for _ in range(a.sessions)]
    split = (a.split_probe != 0) and not a.direct     # host-gather tier only: the device-driven tier is faster fused
    for s in sessions:
        s.set_option("timing", 1)
        s.set_option("probe_variant", a.unroll)
        s.set_option("split_probe", 1 if split else 0)

    # resident set = what the warm-up actually placed (first C rows in file order minus over-full buckets)
    C = int(np.ceil(a.cache_frac * R))
    resident = []
    for t in range(T):
        k = np.arange(C, dtype=np.int64)
        resident.append(k[cache.query(t, k) >= 0])
    resident_frac = float(np.mean([r.size / C for r in resident]))

    gen = torch.Generator(device="cuda")
    gen.manual_seed(SEED + rank)
    cdf_d = torch.from_numpy(zipf_cdf(C, a.zipf)).cuda()
    resident_d = [torch.from_numpy(r).cuda() for r in resident]
    nb = min(a.distinct_batches, a.steps + a.warmup) if a.distinct_batches > 0 else a.steps + a.warmup
    batches_d = make_batches_gpu(torch, gen, resident_d, cdf_d, R, C, B, a.hit, nb)
    batches_h = [b.cpu().numpy() for b in batches_d[: min(8, nb)]]
    del cdf_d, resident_d
    outs = [torch.empty(N * D, dtype=torch.float32, device="cuda") for _ in sessions]
    nk = [B] * T
    torch.cuda.synchronize()

    ncpu = effective_cpus()
    lat_ms, kern_ms, miss_ct, phases, uniq_ct, gpu_ms, gath_ms = [], [], [], [], [], [], []
    lock = threading.Lock()
    post_hooks = []   # per session: work appended to every step (the config-5 leg runs the dense step here)
    step_hooks = []   # per session: replaces the step

    def run_steps(count, record, first=0):
        nxt = [0]

        def worker(si):
            s = sessions[si]
            while True:
                with lock:
                    i = nxt[0]
                    if i >= count:
                        return
                    nxt[0] += 1
                t0 = time.perf_counter()
                if step_hooks:     # a leg that replaces the whole step (the fused lookup+interaction call)
                    step_hooks[si](si, batches_d[(first + i) % len(batches_d)])
                else:
                    s.lookup_device(batches_d[(first + i) % len(batches_d)], nk, out=outs[si])
                if post_hooks:
                    post_hooks[si](si)
                dt = (time.perf_counter() - t0) * 1e3
                st = s.last_stats()
                if record:
                    with lock:
                        lat_ms.append(dt)
                        kern_ms.append(st.probe_gather_ms)
                        gath_ms.append(st.hit_gather_ms)
                        miss_ct.append(st.misses)
                        uniq_ct.append(st.unique_misses)
                        gpu_ms.append(st.gpu_call_ms)
                        phases.append([float(x) for x in st.phase_ms])

        th = [threading.Thread(target=worker, args=(si,)) for si in range(len(sessions))]
        [x.start() for x in th]
        [x.join() for x in th]

    run_steps(a.warmup, False)
    if a.mode == "async":
        cache.wait_async()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    thr0 = cpu_throttle_stat()
    t0 = time.perf_counter()
    run_steps(a.steps, True, first=a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    thr1 = cpu_throttle_stat()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- extra legs (outside the timed region; per-GPU numbers of this rank) --------------------------------
    extra = {}
    main_lat, main_kern, main_miss, main_phases = list(lat_ms), list(kern_ms), list(miss_ct), list(phases)
    main_uniq = list(uniq_ct)
    main_gath = list(gath_ms)
    for s in sessions:   # the extra legs measure the fused kernel (their kernel times are K_A's)
        s.set_option("split_probe", 0)
    main_gpu = list(gpu_ms)
    if not a.no_extra_legs and world == 1:  # informational legs: single-GPU run only
        def leg(batches, steps, sess_list):
            lat_ms.clear(); kern_ms.clear(); miss_ct.clear(); phases.clear()
            saved = sessions[:]
            sessions[:] = sess_list
            batches_d_saved = batches_d[:]
            batches_d[:] = batches
            run_steps(4, False)
            torch.cuda.synchronize()
            tl0 = time.perf_counter()
            run_steps(steps, True, first=4)
            torch.cuda.synchronize()
            dt = time.perf_counter() - tl0
            sessions[:] = saved
            batches_d[:] = batches_d_saved
            k = float(np.mean(kern_ms))
            return {"lookups_per_s": steps * N / dt, "ms_per_step": dt / steps * 1e3, "avg_kernel_ms": k,
                    "kernel_frac_of_hbm_peak": N * (8 + 8 * D) / (k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "measured_hit_rate": 1.0 - float(np.mean(miss_ct)) / N}

        gen2 = torch.Generator(device="cuda")
        gen2.manual_seed(SEED + 1000 + rank)
        cdf_d = torch.from_numpy(zipf_cdf(C, a.zipf)).cuda()
        resident_d = [torch.from_numpy(r).cuda() for r in resident]
        def run_legs():
            # (1) every key resident: the GPU-side ceiling of the path, one session (kernel runs alone)
            hot_batches = make_batches_gpu(torch, gen2, resident_d, cdf_d, R, C, B, 1.1, 8)
            extra["all_hit_one_session"] = leg(hot_batches, 24, sessions[:1])
            # (2) the reference's default policy at this hit rate (hit_rate_threshold 0.9 < 0.95): missed keys
            #     return the default vector now and are fetched + inserted in the background
            fresh = make_batches_gpu(torch, gen2, resident_d, cdf_d, R, C, B, a.hit, 28)
            for s in sessions:
                s.set_option("hit_rate_threshold_permille", 900)
            extra["async_insert_threshold_0.9"] = leg(fresh, 24, sessions)
            cache.wait_async()
            for s in sessions:
                s.set_option("hit_rate_threshold_permille", 1000 if a.mode == "sync" else 500)
            # (3) BASELINE config 5: the dense step (bottom MLP 13-512-256-D + dot interaction, fp16 MFMA) consuming
            #     OUTPUT0 where the lookup left it.  Kernel time alone, then lookup + dense per step with both sessions.
            if D % 32 == 0 and D <= 512 and T <= 31:
                from hugectr_backend_amd.dense import DenseInteraction
                rngw = np.random.default_rng(SEED)
                dims, k = [512, 256, D], 13
                ws, bs = [], []
                for n in dims:
                    ws.append(((rngw.random((k, n), dtype=np.float32) * 2 - 1) * (1.5 / np.sqrt(k))).astype(np.float32))
                    bs.append(((rngw.random(n, dtype=np.float32) - 0.3) * 0.2).astype(np.float32))
                    k = n
                ops = [DenseInteraction(ws, bs, T, D, device=dev) for _ in sessions]
                xd = torch.randn(B, 13, device="cuda")
                outd = [torch.empty((B, ops[0].out_stride), dtype=torch.float16, device="cuda") for _ in sessions]
                for _ in range(3):
                    ops[0].forward(xd, outs[0], B, out=outd[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    ops[0].forward(xd, outs[0], B, out=outd[0])
                e1.record()
                torch.cuda.synchronize()
                dense_ms = e0.elapsed_time(e1) / 20
                dense_bytes = N * 4 * D + B * 13 * 4 + 2 * B * D * 2 + B * ops[0].out_stride * 2
                dense_flops = 2 * B * (16 * 512 + 512 * 256 + 256 * D) + 2 * B * 32 * 32 * D
                post_hooks[:] = [lambda si: (ops[si].forward(xd, outs[si], B, out=outd[si]),
                                             torch.cuda.current_stream().synchronize()) for _ in sessions]
                fresh5 = make_batches_gpu(torch, gen2, resident_d, cdf_d, R, C, B, a.hit, 28)
                c5 = leg(fresh5, 24, sessions)
                post_hooks[:] = []
                c5.update({"dense_kernels_ms": dense_ms, "dense_algorithmic_bytes": dense_bytes,
                           "dense_frac_of_hbm_peak": dense_bytes / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "dense_mfma_tflops": dense_flops / (dense_ms * 1e-3) / 1e12,
                           "samples_per_s": c5["lookups_per_s"] / T,
                           "note": "lookup (sync insert, exact rows) + bottom MLP 13-512-256-%d + dot interaction per step; "
                                   "output [batch, %d] f16" % (D, ops[0].out_dim)})
                extra["c5_lookup_plus_dense"] = c5
                del fresh5
                # (3b) the same step with the lookup fused into the interaction: probe only, rows read from the cache
                #      slots / miss staging by the interaction kernel, OUTPUT0 never written (device-driven tier only)
                if a.direct and a.mode == "sync":
                    fresh5b = make_batches_gpu(torch, gen2, resident_d, cdf_d, R, C, B, a.hit, 28)
                    step_hooks[:] = [lambda si, keys: ops[si].lookup_interact(sessions[si], keys, B, xd, out=outd[si]) for _ in sessions]
                    c5f = leg(fresh5b, 24, sessions)
                    step_hooks[:] = []
                    c5f["probe_only_kernel_ms"] = c5f.pop("avg_kernel_ms")   # the probe moves no rows here:
                    c5f.pop("kernel_frac_of_hbm_peak")                        # the gather roofline does not apply to it
                    c5f.update({"samples_per_s": c5f["lookups_per_s"] / T,
                                "note": "one call per step: probe, miss fetch, bottom MLP, interaction reading cache slots / staging, insert"})
                    extra["c5_fused_lookup_interact"] = c5f
                    del fresh5b
            # (4) the miss path arranged as in the reference (host threads gather the missed rows, hipMemcpyAsync ships
            #     them) on the very same cache and tables: session option "host_gather"
            if a.direct and a.mode == "sync":
                fresh6 = make_batches_gpu(torch, gen2, resident_d, cdf_d, R, C, B, a.hit, 28)
                for s in sessions:
                    s.set_option("host_gather", 1)
                extra["host_gather_tier_same_cache"] = leg(fresh6, 24, sessions)
                for s in sessions:
                    s.set_option("host_gather", 0)
                del fresh6
            # (5) the headline ran with the split probe: the same workload with the fused probe+gather kernel
            if split and a.mode == "sync":
                fresh7 = make_batches_gpu(torch, gen2, resident_d, cdf_d, R, C, B, a.hit, 28)
                extra["fused_probe_gather_same_cache"] = leg(fresh7, 24, sessions)
                del fresh7

        try:   # the legs are informational: a failure in one of them must not cost the headline line
            run_legs()
        except Exception as e:  # noqa: BLE001
            extra["legs_error"] = repr(e)[:300]
            sys.stderr.write(f"[bench] extra legs stopped: {e!r}\n")
        finally:
            post_hooks[:] = []
            step_hooks[:] = []
            for s in sessions:
                s.set_option("host_gather", 0)
                s.set_option("hit_rate_threshold_permille", 1000 if a.mode == "sync" else 500)
        del cdf_d, resident_d

    # ---- untimed parity check of the last step of session 0 against the CPU oracle (tables 0..1) ----
    parity = None
    cpu = None
    checker_note = None
    if rank == 0:
        def run_checker():
            nonlocal parity, cpu
            from oracle import hps_oracle as O
            chk_tables = min(2, T)
            # which batch did session 0 run last?  re-run one known batch to be sure
            sessions[0].lookup_device(batches_d[0], nk, out=outs[0])
            torch.cuda.synchronize()
            got = outs[0][: chk_tables * B * D].cpu().numpy()
            co = O.COracle()
            sample_rows = []
            for t in range(chk_tables):
                rows = np.empty((R, D), dtype=np.float32)
                # generate the oracle's copy of the table in parallel slabs (C code releases the GIL)
                nth = ncpu
                step = (R + nth - 1) // nth

                def gen(lo, t=t, rows=rows):
                    hi = min(R, lo + step)
                    if hi > lo:
                        O.c_synth_rows(SEED, t, lo, hi - lo, D, out=rows[lo:hi])

                th = [threading.Thread(target=gen, args=(lo,)) for lo in range(0, R, step)]
                [x.start() for x in th]
                [x.join() for x in th]
                sample_rows.append(rows)
            keys_seq = np.arange(R, dtype=np.int64)
            for t in range(chk_tables):
                co.add_table_arrays(keys_seq, sample_rows[t])
            q = batches_h[0][: chk_tables * B]
            ref = co.lookup(q, [B] * chk_tables, [0.0] * chk_tables, threads=ncpu)
            if a.mode == "sync":
                parity = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
            else:
                # async mode: resident keys exact, others default
                same = got.view(np.uint32).reshape(-1, D) == ref.view(np.uint32).reshape(-1, D)
                is_default = (got.reshape(-1, D) == 0.0).all(axis=1)
                parity = bool((same.all(axis=1) | is_default).all())

            if not a.no_cpu_baseline and world == 1:  # timed on rank 0 at N=1 only (the other ranks' pools share the cores)
                # ---- cpu_baseline: the oracle (a port of the reference's hash_map parameter-server lookup)
                # on the same key batches, tables 0..chk_tables-1 only (bounded sample), all host cores ----
                threads = ncpu  # the CPUs the container may use (cgroup quota), not the visible hardware threads
                nkc = [B] * chk_tables
                outc = np.empty(chk_tables * B * D, dtype=np.float32)
                done, tc0 = 0, time.perf_counter()
                reps = 0
                while time.perf_counter() - tc0 < a.cpu_seconds and reps < 2000:
                    qb = batches_h[reps % len(batches_h)][: chk_tables * B]
                    co.lookup(qb, nkc, [0.0] * chk_tables, threads=threads, out=outc)
                    done += qb.size
                    reps += 1
                tcpu = time.perf_counter() - tc0
                cpu = {
                    "value": done / tcpu, "unit": "lookups/s", "cores": threads, "kind": "port",
                    "sample": f"{reps} passes over the first {chk_tables} of {T} tables' key slices "
                              f"({chk_tables * B} keys/pass, {R} rows x {D} fp32 per table), oracle/hps_oracle.c "
                              f"oracle_lookup_mt with {threads} threads",
                }

        try:   # the oracle is the checker; if it cannot run here the measurement still stands, marked unchecked
            run_checker()
        except Exception as e:  # noqa: BLE001
            checker_note = repr(e)[:300]
            sys.stderr.write(f"[bench] oracle check / cpu baseline stopped: {e!r}\n")

    if rank == 0:
        lat_ms, kern_ms, miss_ct, phases = main_lat, main_kern, main_miss, main_phases
        # HBM traffic of the kernel comes from the committed rocprofv3 PMC passes (bench.py cannot run the
        # profiler on itself): profiles/pmc_latest.json, written by tools/summarize_profile.py
        traffic, traffic_src = None, None
        try:
            pj = json.loads((ROOT / "profiles" / "pmc_latest.json").read_text())
            if pj.get("workload_keys") == N and pj.get("dim") == D:
                traffic = pj["pmc"]["hbm_bytes_per_launch_fetch_doubled"]
                traffic_src = pj.get("source")
        except Exception:
            pass
        k_ms = float(np.mean(kern_ms)) if kern_ms else float("nan")
        fetch_ms = float(np.mean(np.array(phases)[:, 1])) if phases else 0.0
        alg_bytes = N * (8 + 8 * D)  # 8 B key + 4D row read + 4D row write per lookup
        hits = N - (float(np.mean(miss_ct)) if miss_ct else 0.0)
        g_ms = float(np.mean([g for g in main_gath if g > 0])) if any(g > 0 for g in main_gath) else 0.0
        split_run = split and g_ms > 0
        probe_ms = k_ms
        roof_kernel = "hps_probe_gather_kernel"
        if split_run:
            # Split probe: the rows are moved by hps_gather_hits_kernel (the dominant, HBM-bound kernel of the call);
            # K_A only probes.  Its algorithmic bytes: 4 B slot word per key + 4D read + 4D write per HIT.
            # The pair against SURVEY 8(d)'s 1,032 B per lookup is reported next to it (frac_probe_plus_gather).
            roof_kernel = "hps_gather_hits_kernel"
            k_ms = g_ms
            alg_bytes = int(N * 4 + hits * 8 * D)
            traffic, traffic_src = None, None
            try:   # the PMC passes of the same profile set, restricted to this kernel
                if pj.get("workload_keys") == N and pj.get("dim") == D:
                    traffic = pj["pmc_by_kernel"]["hps_gather_hits"]["hbm_bytes_per_launch_fetch_doubled"]
                    traffic_src = pj.get("source")
            except Exception:
                pass
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        value = world * a.steps * N / elapsed
        res = {
            "metric": "embedding lookups/sec, Criteo 26-slot 64K batch",
            "value": value,
            "unit": "lookups/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp32 rows / int64 keys (moved, never computed)",
            "data": "synthetic",
            "config": {
                "workload": f"Criteo DLRM {T} sparse slots, {R} rows/table"
                            + (f" (requested {rows_requested}; reduced to fit the host memory / setup time of {world} replicas{setup_note})" if R != rows_requested else "")
                            + f" x {D}-dim, {B} batch ({N} keys), "
                            f"gpucacheper {a.cache_frac}, 95% cache hit (resident-draw probability {a.hit}; see measured_hit_rate), "
                            f"zipf {a.zipf} within the resident set, "
                            f"{a.mode} insert, {a.sessions} lookup sessions, keys resident in HBM",
                "parallelism": "replicas" if world > 1 else "single",
                "ps_tier": "device-driven (ps_direct_access)" if a.direct else
                           ("host gather" + (f" [{direct_note}]" if direct_note else "")),
            },
            "p50_batch_latency_ms": float(np.percentile(lat_ms, 50)) if lat_ms else None,
            "p99_batch_latency_ms": float(np.percentile(lat_ms, 99)) if lat_ms else None,
            # GPU side of a batch (HIP events on the session's stream: probe+gather start to the last kernel of the call)
            "p50_batch_gpu_ms": float(np.percentile(main_gpu, 50)) if main_gpu else None,
            "p99_batch_gpu_ms": float(np.percentile(main_gpu, 99)) if main_gpu else None,
            "measured_hit_rate": 1.0 - float(np.mean(miss_ct)) / N if miss_ct else None,
            # the five slowest calls of the timed region: [end-to-end ms, ms until the miss counts are on the host,
            # ms inside the host gather calls, ms of upload tail + scatter + insert, ms of the whole call inside the engine]
            # -- says which wait a multi-millisecond stall sat in
            "slowest_calls_ms": [[round(l, 3)] + [round(x, 3) for x in ph]
                                 for l, ph in sorted(zip(lat_ms, phases), key=lambda t: -t[0])[:5]] if lat_ms and phases else None,
            # host side of the timed region: CPUs this process may use, and how often the cgroup's CPU quota stopped it
            "host": {"cpus": ncpu,
                     "cpu_quota_throttled_periods_in_timed_region": (thr1[0] - thr0[0]) if thr0 and thr1 else None,
                     "cpu_quota_throttled_ms_in_timed_region": (thr1[1] - thr0[1]) / 1e3 if thr0 and thr1 else None},
            "resident_fraction_after_warmup": resident_frac,
            "roofline": {
                "bound": "hbm",
                "kernel": roof_kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_kernel_ms": k_ms,
                # `frac` is measured inside the timed region, where the kernel shares the chip with the other
                # session's PCIe fetch / dedup / insert kernels; the same kernel with nothing underneath
                # (all-hit leg, one session) is reported next to it
                "frac_kernel_alone": (extra.get("all_hit_one_session") or {}).get("kernel_frac_of_hbm_peak"),
                # the same kernel, same cache, same workload and session count with the miss path arranged as in
                # the reference (host gather + hipMemcpyAsync: the DMA engine does not disturb it; the job is slower)
                "frac_with_host_gather_tier": (extra.get("host_gather_tier_same_cache") or {}).get("kernel_frac_of_hbm_peak"),
                # the read side alone, and both against the measured copy ceiling of the part
                "read_only_frac": ((N * 4 + hits * 4 * D) if split_run else N * (8 + 4 * D)) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None,
                "frac_of_copy_ceiling_6290": achieved / 6290.0,
                # SURVEY.md 8(d) prices every lookup at 8 + 4D + 4D bytes; the kernel itself moves rows only for the
                # keys that hit (a missed key's row is written later by the scatter kernel).  Bytes the kernel really
                # has to move = 8 per key + 8D per hit: the stricter figure, and the one to read at low hit rates
                "frac_hit_rows_only": (((N * 4 if split_run else N * 8) + hits * 8 * D) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                       if k_ms > 0 and miss_ct else None),
                # split probe: the probe kernel's time, the pair priced at SURVEY 8(d)'s 1,032 B per lookup over both
                # kernels' time, and the fused kernel on the same cache and workload (leg fused_probe_gather_same_cache)
                "probe_kernel_ms": probe_ms if split_run else None,
                "frac_probe_plus_gather": (N * (8 + 8 * D) / ((probe_ms + g_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS if split_run else None),
                "frac_fused_kernel": (extra.get("fused_probe_gather_same_cache") or {}).get("kernel_frac_of_hbm_peak"),
            },
            # the other leg of the synchronous path: the missed rows cross PCIe once each.  Device-driven tier:
            # HIP-event time of hps_ps_fetch_direct_kernel; bytes = unique missed rows x 4*D.
            "roofline_pcie": ({
                "bound": "pcie", "kernel": "hps_ps_fetch_direct_kernel",
                "achieved": float(np.mean(main_uniq)) * 4 * D / (fetch_ms * 1e-3) / 1e9,
                "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                "frac": float(np.mean(main_uniq)) * 4 * D / (fetch_ms * 1e-3) / 1e9 / PCIE_PEAK_GBS,
                "avg_kernel_ms": fetch_ms, "unique_missed_rows_per_batch": float(np.mean(main_uniq)),
            } if a.direct and fetch_ms > 0 else None),
            # [1] = wall time of the host gather (host tier) or GPU time of the fetch kernel (device-driven tier)
            "mean_phase_ms": dict(zip(["probe_gather_dedup_until_counts", "ps_fetch", "h2d_scatter_insert", "call"],
                                      [float(x) for x in np.mean(np.array(phases), axis=0)])) if phases else None,
            "extra_legs": extra or None,
            "cpu_baseline": cpu,
            "parity_vs_oracle_bit_exact": parity,
            "checker_note": checker_note,
            "setup_seconds": {"host_tables": t_tables, "gpu_cache_warmup": t_cache},
            "cache_counters": cache.counters(),
        }
    else:
        res = None
    for s in sessions:
        s.close()

    # ---- one GPU, headline measured on the host-gather tier: the device-driven tier (ps_direct_access) on the same
    #      workload right after, with the headline's resources released first (its tables are page-locked: a second
    #      133 GB next to the first would not fit the box) ----
    if world == 1 and not a.direct and not a.no_extra_legs and not a.no_direct_leg:
        import gc
        del sessions, cache, ps, made
        gc.collect()
        torch.cuda.empty_cache()
        try:
            dleg = direct_tier_leg(a, torch, hps, T, R, D, B, N, dev, outs, cfg)
        except Exception as e:  # noqa: BLE001
            dleg = {"error": repr(e)[:300]}
            sys.stderr.write(f"[bench] device-driven tier leg stopped: {e!r}\n")
        fused = dleg.pop("c5_fused_lookup_interact", None) if isinstance(dleg, dict) else None
        res["extra_legs"] = dict(res["extra_legs"] or {}, device_driven_tier=dleg)
        if fused:
            res["extra_legs"]["c5_fused_lookup_interact"] = fused
        res["roofline"]["frac_under_device_driven_tier"] = dleg.get("kernel_frac_of_hbm_peak")

    # ---- BASELINE config 3 leg (N > 1 only): ONE table sharded over the ranks, RCCL all-to-all of keys and rows ----
    # Runs after the headline measurement is complete and its resources are released; a watchdog prints the headline
    # line and ends every rank if the leg does not finish (a collective that hangs cannot be caught any other way).
    if world > 1 and not a.no_sharded_leg:
        import gc
        del sessions, outs, batches_d, cache, ps, made
        gc.collect()
        torch.cuda.empty_cache()

        def give_up():
            if rank == 0:
                res.setdefault("extra_legs", None)
                res["extra_legs"] = dict(res["extra_legs"] or {}, sharded_c3={"error": f"no result within {a.sharded_timeout} s"})
                print(json.dumps(res), flush=True)
            os._exit(0)

        dog = threading.Timer(a.sharded_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            leg = sharded_leg(a, torch, dist, hps, rank, world, local_rank, coll_dev)
        except Exception as e:  # noqa: BLE001
            leg = {"error": repr(e)[:300]}
        dog.cancel()
        if rank == 0:
            res["extra_legs"] = dict(res["extra_legs"] or {}, sharded_c3=leg)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def direct_tier_leg(a, torch, hps, T, R, D, B, N, dev, outs, cfg):
    """The headline workload on a ps_direct_access deployment of the same model (own server: page-locked tables, device
    index): two sessions, fresh batches, exact rows; then the fused lookup+interaction call of config 5."""
    model = cfg["models"][0]["model"]
    cfg = json.loads(json.dumps(cfg))
    cfg["models"][0]["ps_direct_access"] = True
    t0 = time.time()
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t in range(T):
        ps.load_table_synthetic(model, t, SEED, 0, R)
    ps.create_embedding_cache_per_model(model)
    cache = ps.get_embedding_cache(model, dev)
    t_setup = time.time() - t0
    sessions = [hps.LookupSession.create(ps, model, cache) for _ in range(a.sessions)]
    split = False   # the device-driven tier keeps the fused kernel (measured: 1.70 split vs 1.85 G lookups/s fused)
    for s in sessions:
        s.set_option("timing", 1)
        s.set_option("probe_variant", a.unroll)
        s.set_option("split_probe", 1 if split else 0)
    C = int(np.ceil(a.cache_frac * R))
    resident = []
    for t in range(T):
        k = np.arange(C, dtype=np.int64)
        resident.append(k[cache.query(t, k) >= 0])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(SEED + 555)
    cdf_d = torch.from_numpy(zipf_cdf(C, a.zipf)).cuda()
    resident_d = [torch.from_numpy(r).cuda() for r in resident]
    nk = [B] * T
    lock = threading.Lock()

    gath = []

    def run(batches, count, first, record, step=None):
        nxt = [0]
        lat, kern, fetch, miss, uniq, gpu = [], [], [], [], [], []
        gath.clear()

        def worker(si):
            s = sessions[si]
            while True:
                with lock:
                    i = nxt[0]
                    if i >= count:
                        return
                    nxt[0] += 1
                keys = batches[(first + i) % len(batches)]
                ts = time.perf_counter()
                if step:
                    step(si, keys)
                else:
                    s.lookup_device(keys, nk, out=outs[si])
                dt = (time.perf_counter() - ts) * 1e3
                st = s.last_stats()
                if record:
                    with lock:
                        # split call: the HBM kernel of the call is the hit gather (the probe moved no rows)
                        lat.append(dt); kern.append(st.hit_gather_ms if st.hit_gather_ms > 0 else st.probe_gather_ms)
                        fetch.append(st.phase_ms[1]); gath.append(st.hit_gather_ms)
                        miss.append(st.misses); uniq.append(st.unique_misses); gpu.append(st.gpu_call_ms)

        th = [threading.Thread(target=worker, args=(si,)) for si in range(len(sessions))]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        torch.cuda.synchronize()
        return time.perf_counter() - t1, lat, kern, fetch, miss, uniq, gpu

    steps = 40
    batches = make_batches_gpu(torch, gen, resident_d, cdf_d, R, C, B, a.hit, steps + 8)
    run(batches, 8, 0, False)
    dt, lat, kern, fetch, miss, uniq, gpu = run(batches, steps, 8, True)
    k_ms, f_ms = float(np.mean(kern)), float(np.mean(fetch))
    split_run = split and any(g > 0 for g in gath)
    hits = N - float(np.mean(miss))
    kern_bytes = (N * 4 + hits * 8 * D) if split_run else N * (8 + 8 * D)
    out = {
        "lookups_per_s": steps * N / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "sessions": len(sessions),
        "p50_batch_latency_ms": float(np.percentile(lat, 50)), "p99_batch_latency_ms": float(np.percentile(lat, 99)),
        "p50_batch_gpu_ms": float(np.percentile(gpu, 50)),
        "measured_hit_rate": 1.0 - float(np.mean(miss)) / N,
        "hbm_kernel": "hps_gather_hits_kernel" if split_run else "hps_probe_gather_kernel",
        "avg_kernel_ms": k_ms, "kernel_frac_of_hbm_peak": kern_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "roofline_pcie": {"bound": "pcie", "kernel": "hps_ps_fetch_direct_kernel", "avg_kernel_ms": f_ms,
                          "achieved": float(np.mean(uniq)) * 4 * D / (f_ms * 1e-3) / 1e9 if f_ms > 0 else None,
                          "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                          "frac": float(np.mean(uniq)) * 4 * D / (f_ms * 1e-3) / 1e9 / PCIE_PEAK_GBS if f_ms > 0 else None},
        "setup_seconds": t_setup,
        "note": "same workload, two sessions, exact rows; the GPU resolves the misses through a device index of the page-locked "
                "host tables and reads the rows over PCIe itself (no host threads on the path)",
    }
    del batches
    if D % 32 == 0 and D <= 512 and T <= 31 and a.mode == "sync":
        from hugectr_backend_amd.dense import DenseInteraction
        rngw = np.random.default_rng(SEED)
        dims, k = [512, 256, D], 13
        ws, bs = [], []
        for n in dims:
            ws.append(((rngw.random((k, n), dtype=np.float32) * 2 - 1) * (1.5 / np.sqrt(k))).astype(np.float32))
            bs.append(((rngw.random(n, dtype=np.float32) - 0.3) * 0.2).astype(np.float32))
            k = n
        ops = [DenseInteraction(ws, bs, T, D, device=dev) for _ in sessions]
        xd = torch.randn(B, 13, device="cuda")
        outd = [torch.empty((B, ops[0].out_stride), dtype=torch.float16, device="cuda") for _ in sessions]
        fb = make_batches_gpu(torch, gen, resident_d, cdf_d, R, C, B, a.hit, 28)
        step = lambda si, keys: ops[si].lookup_interact(sessions[si], keys, B, xd, out=outd[si])  # noqa: E731
        run(fb, 4, 0, False, step)
        dtf, latf, kernf, _, missf, _, _ = run(fb, 24, 4, True, step)
        out["c5_fused_lookup_interact"] = {
            "lookups_per_s": 24 * N / dtf, "ms_per_step": dtf / 24 * 1e3, "samples_per_s": 24 * B / dtf,
            "measured_hit_rate": 1.0 - float(np.mean(missf)) / N, "probe_only_kernel_ms": float(np.mean(kernf)),
            "note": "one call per step: probe, miss fetch, bottom MLP, interaction reading cache slots / staging, insert",
        }
    for s in sessions:
        s.close()
    return out


def sharded_leg(a, torch, dist, hps, rank, world, local_rank, coll_dev):
    """One table of Rt rows x D sharded over the P ranks (owner = mix64(key) mod P), every rank resident at 100 % in
    its own HBM; per step every rank looks up N/P uniform keys: counting sort by owner, all-to-all keys, local
    lookup, all-to-all rows, un-permute (hugectr_backend_amd/sharded.py).  Global lookups/s = N x steps / time."""
    from hugectr_backend_amd.sharded import ShardedLookup
    P, D = world, a.dim
    N = a.tables * a.batch
    n_local = N // P
    budget = int(host_memory_budget() * 0.6)          # this rank's share is checked against the minimum over ranks
    bt = torch.tensor([budget], dtype=torch.int64, device=coll_dev)
    dist.all_reduce(bt, op=dist.ReduceOp.MIN)
    per_rank_rows = min(a.shard_rows // P, int(bt.item()) // P // (4 * D + 80))
    Rt = per_rank_rows * P
    model = "criteo_sharded"
    recv_cap = int(n_local * 1.25) + 4096
    cfg = {
        "supportlonglong": True,
        "volatile_db": {"type": "hash_map", "num_partitions": 8},
        "models": [{
            "model": model, "sparse_files": ["synthetic://shard"], "num_of_worker_buffer_in_pool": 2,
            "embedding_vecsize_per_table": [D], "maxnum_catfeature_query_per_table_per_sample": [1],
            "default_value_for_each_table": [0.0], "deployed_device_list": [local_rank], "max_batch_size": recv_cap,
            "gpucache": True, "gpucacheper": 1.0, "hit_rate_threshold": 1.0, "ps_direct_access": bool(a.direct),
        }],
    }
    t0 = time.time()
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    ps.load_table_synthetic(model, 0, SEED, 0, Rt, shard=rank, num_shards=P)
    ps.create_embedding_cache_per_model(model)
    cache = ps.get_embedding_cache(model, local_rank)
    sess = hps.LookupSession.create(ps, model, cache)
    sl = ShardedLookup(sess)
    t_setup = time.time() - t0
    gen = torch.Generator(device="cuda")
    gen.manual_seed(SEED + 77 + rank)
    steps, warm = a.sharded_steps, 5
    batches = [torch.randint(0, Rt, (n_local,), generator=gen, device="cuda", dtype=torch.int64) for _ in range(8)]
    for i in range(warm):
        out = sl.lookup(batches[i % 8])
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    lat = []
    for i in range(steps):
        ts = time.perf_counter()
        out = sl.lookup(batches[i % 8])
        torch.cuda.current_stream().synchronize()
        lat.append((time.perf_counter() - ts) * 1e3)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    # parity of this rank's last answer against the CPU oracle's row recipe (256 sampled positions)
    from oracle import hps_oracle as O
    kh = batches[(steps - 1) % 8].cpu().numpy()
    got = out.view(-1, D)
    idx = np.linspace(0, n_local - 1, 256).astype(np.int64)
    exp = np.concatenate([O.c_synth_rows(SEED, 0, int(kh[i]), 1, D) for i in idx]).reshape(256, D)
    ok = bool(np.array_equal(got[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32), exp.view(np.uint32)))
    okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=coll_dev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    sent_remote = float(sum(c for r, c in enumerate(sl.last_sent) if r != rank))
    res = {
        "workload": f"one table of {Rt} rows x {D} fp32 sharded over {P} ranks by mix64(key) mod {P}, 100 % resident in HBM "
                    f"({per_rank_rows} rows per rank), {N} uniform keys per step in total ({n_local} issued per rank)",
        "lookups_per_s": N * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
        "p50_step_latency_ms": float(np.percentile(lat, 50)),
        "rows_bytes_sent_per_rank_per_step": sent_remote * 4 * D,
        "row_exchange_GBps_per_rank": sent_remote * 4 * D / (dt / steps) / 1e9,
        "xgmi_note": "row all-to-all is bounded by 7 links x ~153 GB/s per GPU (MI355X_MICROARCH.md); the figure above "
                     "divides by the WHOLE step time, not the collective's own",
        "parity_vs_oracle_bit_exact": bool(okt.item()), "backend": dist.get_backend(), "setup_seconds": t_setup,
    }
    sess.close()
    return res


if __name__ == "__main__":
    main()
