#!/usr/bin/env python3
"""Headline benchmark: embedding lookups/s on Criteo-shaped key batches (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one call of the reference's lookup contract — LookupSession::lookup(h_keys_per_table, d_vectors_per_table,
num_keys_per_table), docs/architecture.md:308-323 — through the C ABI (hps_session_lookup) on one batch of synthetic keys
(26 tables x 65,536 keys) that sit in ordinary HOST memory, as a Triton request's KEYS do (hps.cc:586-597): key staging +
upload (narrowed to 3 or 4 bytes per key when every key fits), tile-local input dedup + cache probe (HIP), call-wide unique misses
(HIP), hit-row gather (HIP) running while the parameter server fetches the missed rows (one GPU, >= 12 CPUs: host threads
gather them and hipMemcpyAsync ships them, as in the reference; otherwise / --direct 1: the device-driven tier, a HIP
kernel reading them out of pinned host memory), scatter + cache insert (HIP).  Output rows land in HBM (Triton's device
output buffer).  Results are the exact fp32 rows (sync-insert mode, hit_rate_threshold=1.0), checked against the CPU
oracle on every run.

Timed region: W warm-up steps, then blocks of exactly K steps, each block bracketed by a barrier and a device
synchronisation on both sides (max over ranks); `value` is K x N / the MEDIAN block, all block times are printed.

N>1 (launched by torch.distributed.run, one rank per GPU): the reference's multi-GPU mode is "replicas only"
(independent cache per GPU) — each rank serves its own batches, no data-path collective; RCCL is used
for the barriers and the max-over-ranks time only.  scaling = weak.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
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
    ap.add_argument("--blocks", type=int, default=0,
                    help="timed blocks of --steps steps each; 0 (default): ceil(240 / steps), at most 12 (12 blocks for --steps 20)")
    ap.add_argument("--tables", type=int, default=26)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per table (BASELINE config 2: 1e7)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536, help="samples per batch; one key per table per sample")
    ap.add_argument("--cache-frac", type=float, default=0.2, help="gpucacheper")
    ap.add_argument("--hit", type=float, default=0.957,
                    help="probability that a key is drawn from the set resident after warm-up.  The MEASURED hit rate is what the "
                         "line reports (measured_hit_rate, config.measured_hit_rate): 0.957 here measures ~0.957 since the admission "
                         "rule of round 4 (before it, evictions of the hot tail brought it to 0.95); the leg hit_950 runs the same two "
                         "sessions at a draw probability tuned to a measured 0.950 +- 0.001")
    ap.add_argument("--zipf", type=float, default=1.05)
    ap.add_argument("--sessions", type=int, default=2, help="concurrent lookup sessions (Triton instance count)")
    ap.add_argument("--mode", choices=["sync", "async"], default="sync",
                    help="sync: exact rows (threshold 1.0); async: misses return default, inserted in background")
    ap.add_argument("--distinct-batches", type=int, default=0, help="0: one fresh batch per step")
    ap.add_argument("--probe-variant", type=int, default=1002, help="probe kernel: 1002 (default) or 1102 (no tile-local input dedup)")
    ap.add_argument("--xcd-walk", type=int, default=1, help="gather kernel: each XCD sweeps its own eighth of the keys")
    ap.add_argument("--probe-in-lane", type=int, default=2, help="K_P and the kernel lane: 1 = always inside, 0 = never, 2 = outside while the session's calls miss little (default)")
    ap.add_argument("--numa-bind", type=int, default=1, help="1: bind this process to the NUMA node of its GPU(s) when they share one (numactl --cpunodebind); 0: leave it alone")
    ap.add_argument("--keys-by-kernel", type=int, default=2, help="staged keys pulled into HBM by a kernel instead of copy-engine copies: 0 never, 1 always, 2 while the session's calls miss much (default)")
    ap.add_argument("--chain-gather", type=int, default=0, help="other sessions' probes wait for a session's gather kernel too")
    ap.add_argument("--narrow-keys", type=int, default=1, help="stage host keys narrower when every key of the request fits: 1 = 3-byte packing or uint32, 2 = uint32 only, 0 = off")
    ap.add_argument("--direct", type=int, default=-1,
                    help="parameter-server tier of the miss path.  0: host threads gather the missed rows and "
                         "hipMemcpyAsync ships them (the reference's arrangement); 1: ps_direct_access (the GPU resolves "
                         "misses itself out of pinned host memory, no host threads); -1 (default): host gather when there are "
                         "at least 12 CPUs per GPU and at most four GPUs, else device-driven — on one GPU the other tier is "
                         "measured right after the headline and reported under extra_legs")
    ap.add_argument("--split-probe", type=int, default=-1,
                    help="1 / -1 (default): the miss counts are read back right behind the probe and the hit rows are gathered "
                         "while the misses are fetched (DESIGN.md 3.4c); 0: gather first")
    ap.add_argument("--no-direct-leg", action="store_true",
                    help="one GPU, host-gather headline: skip the device-driven-tier leg measured afterwards")
    ap.add_argument("--no-wide-leg", action="store_true",
                    help="one GPU: skip the leg with keys offset by 2^40 (8 bytes per key over PCIe) at 95 %% hit")
    ap.add_argument("--no-triton-leg", action="store_true",
                    help="one GPU: skip the leg through TRITONBACKEND_ModelInstanceExecute (tools/triton_abi_bench.cpp)")
    ap.add_argument("--triton-timeout", type=float, default=240.0)
    ap.add_argument("--no-c3-leg", action="store_true",
                    help="one GPU: skip the BASELINE config 3 leg with P = 4 LOGICAL shards in this process (in-process transport)")
    ap.add_argument("--no-sharded-leg", action="store_true",
                    help="N>1: skip the BASELINE config 3 leg (one table sharded over the ranks, RCCL all-to-all)")
    ap.add_argument("--shard-rows", type=int, default=1 << 28,
                    help="rows of the sharded table in total (config 3 names 1e9 = 512 GB; default 2^28 = 137 GB)")
    ap.add_argument("--copy-piece-keys", type=int, default=0, help="config 3, staged_copy transport: keys per piece an owner gathers and ships (0: automatic — 131,072 for a shard on another GPU, one piece for a shard on the entry GPU)")
    ap.add_argument("--selftest-timeout", type=float, default=30.0, help="N>1: deadline of the multi-GPU first-contact self-test (peer access, 4-KB peer stores, per-pair GB/s, one RCCL all-reduce)")
    ap.add_argument("--no-selftest", action="store_true", help="N>1: skip the multi-GPU self-test")
    ap.add_argument("--sharded-steps", type=int, default=50)
    ap.add_argument("--sharded-timeout", type=float, default=180.0, help="bound on each config-3 leg (they run on a thread of their own; the line is printed without a leg that does not come back)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the untimed legs reported next to the headline")
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


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.update(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_local_cpus(pci_bdf, allowed, sysfs="/sys"):
    """CPUs of the NUMA node the GPU hangs off, clipped to `allowed` — or None when the machine has one node, the kernel
    does not know the GPU's node, or too few CPUs would be left.  N replicas on a two-socket node each keep their host
    tables (first touch) and their gather threads next to their own GPU's PCIe root instead of behind the socket link."""
    try:
        nodes = [d for d in os.listdir(os.path.join(sysfs, "devices/system/node")) if d.startswith("node") and d[4:].isdigit()]
        if len(nodes) < 2:
            return None
        dev = os.path.join(sysfs, "bus/pci/devices", pci_bdf)
        if int(open(os.path.join(dev, "numa_node")).read()) < 0:
            return None
        cpus = parse_cpulist(open(os.path.join(dev, "local_cpulist")).read()) & set(allowed)
        return cpus if len(cpus) >= 4 else None
    except Exception:
        return None


def cpu_throttle_stat():
    """(nr_throttled, throttled_usec) of this container's cgroup, or None: a run whose timed region was throttled by the
    CPU quota shows multi-millisecond stalls in its p99 that have nothing to do with the GPU path."""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except Exception:
        return None


def cpu_times():
    """(steal seconds of the machine, CPU seconds this process has used): steal time is what a hypervisor took away from
    the guest's CPUs; the process's CPU seconds per wall second say how close the run is to the CPU budget."""
    steal = None
    try:
        f = open("/proc/stat").readline().split()
        steal = int(f[8]) / os.sysconf("SC_CLK_TCK")
    except Exception:
        pass
    t = os.times()
    return steal, t.user + t.system


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


def plan_replicas(n_rep, ndev, base="criteo_dlrm"):
    """Replica g -> (device, model).  N GPUs: ONE model deployed on devices 0..N-1 (deployed_device_list), one cache per
    device, as the reference does (model_state.cpp:395-420).  Fewer GPUs than replicas (development box): replica g runs on
    device g % ndev, and the j-th extra replica of a device deploys the model again as <base>_dup<j> (the engine keys its
    caches by (model, device)).  Returns (models, model of each replica, {model: its deployed_device_list})."""
    models, rep_model, seen = [], [], {}
    for g in range(n_rep):
        d = g % ndev
        j = seen.get(d, 0)
        seen[d] = j + 1
        name = base if j == 0 else f"{base}_dup{j}"
        rep_model.append(name)
        if name not in models:
            models.append(name)
    deployed = {m: sorted({g % ndev for g in range(n_rep) if rep_model[g] == m}) for m in models}
    return models, rep_model, deployed


class Runner:
    """Drives the lookup sessions of one deployment (server + cache): one host thread per session, each step one call of
    the hot path on one batch, per-step statistics from the engine (HIP events on the session's stream)."""

    MODES = ("host", "pinned", "device")

    def __init__(self, torch, hps, sessions, T, B, D, dev):
        self.torch, self.hps, self.sessions = torch, hps, sessions
        self.T, self.B, self.D, self.N = T, B, D, T * B
        self.outs = [torch.empty(self.N * D, dtype=torch.float32, device=torch.device("cuda", dev)) for _ in sessions]
        self.nk = [B] * T
        self.counts = hps.LookupSession.pack_counts(self.nk)
        self.vptrs = [hps.LookupSession.pack_ptrs([o.data_ptr() + 4 * t * B * D for t in range(T)]) for o in self.outs]
        self.lock = threading.Lock()
        self.post_hooks = []   # per session: appended to every step (config 5: the dense step)
        self.step_hooks = []   # per session: replaces the step

    def pack_host(self, batch_np):
        return self.hps.LookupSession.pack_ptrs([batch_np.ctypes.data + 8 * t * self.B for t in range(self.T)])

    def run(self, batches, count, first=0, mode="host", sess=None, record=None):
        """`count` steps over `batches` starting at index `first`, shared by the sessions in `sess`.
        batches: mode host/pinned -> list of (array, packed pointers); mode device -> list of device tensors."""
        sess = list(range(len(self.sessions))) if sess is None else sess
        nxt = [0]

        def worker(si):
            s = self.sessions[si]
            while True:
                with self.lock:
                    i = nxt[0]
                    if i >= count:
                        return
                    nxt[0] += 1
                b = batches[(first + i) % len(batches)]
                t0 = time.perf_counter()
                if self.step_hooks:
                    self.step_hooks[si](si, b)
                elif mode == "device":
                    s.lookup_device(b, self.nk, out=self.outs[si])
                else:
                    s.lookup_packed(b[1], self.vptrs[si], self.counts)
                if self.post_hooks:
                    self.post_hooks[si](si)
                dt = (time.perf_counter() - t0) * 1e3
                if record is not None:
                    st = s.last_stats()
                    with self.lock:
                        record.append((dt, st.probe_gather_ms, st.hit_gather_ms, st.scatter_ms, st.insert_ms, st.misses,
                                       st.unique_misses, st.gpu_call_ms, [float(x) for x in st.phase_ms], st.key_stage_ms,
                                       st.keys_narrowed, st.key_bytes))

        th = [threading.Thread(target=worker, args=(si,)) for si in sess]
        [x.start() for x in th]
        [x.join() for x in th]


def summarize(rec, N, D, dt, steps):
    """One leg's figures from its per-step records."""
    a = np.array([[r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[9]] for r in rec], dtype=np.float64)
    probe, gather, scatter = float(a[:, 1].mean()), float(a[:, 2].mean()), float(a[:, 3].mean())
    hbm_ms = probe + gather + scatter
    return {
        "lookups_per_s": steps * N / dt, "ms_per_step": dt / steps * 1e3,
        "p50_call_ms": float(np.percentile(a[:, 0], 50)), "p99_call_ms": float(np.percentile(a[:, 0], 99)),
        "max_call_ms": float(a[:, 0].max()),
        "probe_ms": probe, "gather_ms": gather, "scatter_ms": scatter, "insert_ms": float(a[:, 4].mean()),
        "frac_of_hbm_peak_1032B_per_lookup": N * (8 + 8 * D) / (hbm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if hbm_ms > 0 else None,
        "measured_hit_rate": 1.0 - float(a[:, 5].mean()) / N,
        "key_stage_ms": float(a[:, 8].mean()),
        # the call as the engine sees it (steady_clock inside hps_session_lookup*, after the key staging): what is left
        # after the kernels is descriptor upload, count read-back, launches and the final synchronisation
        "engine_call_ms": float(np.mean([r[8][3] for r in rec])),
    }


def _sig(x, nd=4):
    """Numbers of the compact line: `nd` significant digits (the full-precision values are in bench_extra.json)."""
    if isinstance(x, bool) or x is None or isinstance(x, (str, int)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{nd}g}")
    if isinstance(x, (list, tuple)):
        return [_sig(v, nd) for v in x]
    if isinstance(x, dict):
        return {k: _sig(v, nd) for k, v in x.items()}
    return x


COMPACT_LIMIT = 4000   # bytes; the driver keeps the last 8 KB of stdout and parses the LAST line (round 3's 21-KB line was cut)


def compact_line(res, limit=COMPACT_LIMIT):
    """The ONE line the driver parses: the contract's keys, `roofline`, `cpu_baseline` and one or two scalars per leg.
    Everything else of `res` goes to bench_extra.json (emit()).  Guaranteed to serialise to at most `limit` bytes: optional
    groups are dropped from the end until it fits."""
    def g(d, *ks):
        return _dig(d, ks)

    rf = res.get("roofline") or {}
    cfg = res.get("config") or {}
    out = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data", "value_mean", "value_min", "value_max", "slow_blocks")}
    # (the two facts the headline leans on come BEFORE the workload text, which is cut at 300 characters)
    out["config"] = {"measured_hit_rate": cfg.get("measured_hit_rate"), "key_bytes_over_pcie": cfg.get("key_bytes_over_pcie"),
                     "workload": str(cfg.get("workload", ""))[:240], "parallelism": str(cfg.get("parallelism", ""))[:100],
                     "ps_tier": str(cfg.get("ps_tier", ""))[:80], "blocks": cfg.get("blocks"), "value_is": cfg.get("value_is"),
                     "resident_draw_probability": cfg.get("resident_draw_probability")}
    stt = res.get("multi_gpu_selftest")
    if stt:
        flat = lambda m_: [v for row in (m_ or []) for v in row if v is not None]   # noqa: E731
        out["multi_gpu_selftest"] = {
            "devices": stt.get("devices"), "timeout": stt.get("timeout"), "stuck_in": (str(stt["stuck_in"])[:100] if stt.get("stuck_in") else None),
            "error": (str(stt["error"])[:100] if stt.get("error") else None),
            "peer_access_all": (all(v == 1 for v in flat(stt.get("peer_access"))) if stt.get("peer_access") is not None else None),
            "store_4k_all_ok": (all(v == 1 for v in flat(stt.get("store_4k_ok"))) if stt.get("store_4k_ok") is not None else None),
            "rccl_allreduce_ok": _dig(stt, ("rccl_allreduce", "ok")), "rccl_allreduce_ms": _dig(stt, ("rccl_allreduce", "ms")),
            "seconds": stt.get("wall_seconds")}
        out["xgmi_pair_GBps_min"] = stt.get("pair_GBps_min")
        out["xgmi_pair_GBps_median"] = stt.get("pair_GBps_median")
    for k in ("p50_batch_latency_ms", "p99_batch_latency_ms", "measured_hit_rate"):
        out[k] = res.get(k)
    out["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_return_path_kernels", "traffic", "traffic_source",
                                               "algorithmic_bytes_per_call", "probe_ms", "gather_ms", "scatter_ms", "insert_ms",
                                               "insert_on_call_path", "frac_kernels_alone", "box_d2d_copy_GBps")}
    out["roofline"]["kernel"] = str(rf.get("kernel", ""))[:110]
    out["roofline"]["excluded"] = (str(rf["excluded"])[:70] if rf.get("excluded") else None)
    if out["roofline"].get("traffic_source"):
        out["roofline"]["traffic_source"] = str(out["roofline"]["traffic_source"])[:64]
    out["roofline_pcie"] = {"frac": g(res, "roofline_pcie", "frac"), "achieved": g(res, "roofline_pcie", "achieved"),
                            "peak": g(res, "roofline_pcie", "peak"), "unit": "GB/s"}
    cb = res.get("cpu_baseline")
    out["cpu_baseline"] = None if not cb else {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                               "kind": cb.get("kind"), "sample": str(cb.get("sample", ""))[:110]}
    out["parity_vs_oracle_bit_exact"] = res.get("parity_vs_oracle_bit_exact")
    out["parity_full_batch"] = res.get("parity_full_batch_vs_direct_row_index")
    ex = res.get("extra_legs") or {}
    legs = {}

    def put(name, v):
        if v is not None:
            legs[name] = v

    # ---- one or two scalars per leg, the legs the reviews asked about first (what does not fit the limit is cut from the END) ----
    G, M_ = 1e-9, 1e-6
    put("hit_950_Glps", _scale(g(ex, "hit_950_two_sessions_host_keys", "lookups_per_s"), G))
    put("hit_950_measured_hit_rate", g(ex, "hit_950_two_sessions_host_keys", "measured_hit_rate"))
    for tag_, leg_ in (("headline", "headline_under_refresh"), ("hit_999", "hit_999_under_refresh")):
        put(f"{tag_}_under_refresh_Glps", _scale(g(ex, leg_, "lookups_per_s"), G))
        put(f"{tag_}_under_refresh_p50_p99_max_ms", [g(ex, leg_, k_) for k_ in ("p50_call_ms", "p99_call_ms", "max_call_ms")] if g(ex, leg_, "p50_call_ms") is not None else None)
        put(f"{tag_}_under_refresh_refresh_GBps", g(ex, leg_, "refresh_GBps_during_leg"))
    put("unchanged_refresh_row_bytes", g(ex, "unchanged_refresh", "row_bytes"))
    c3e = ex.get("sharded_c3_single_entry") or {}
    put("c3_single_entry_P", c3e.get("shards"))
    put("c3_single_entry_devices", sorted(set(c3e["shard_devices"])) if c3e.get("shard_devices") else None)
    put("c3_single_entry_Glps", _scale(c3e.get("lookups_per_s"), G))
    put("c3_single_entry_p50_ms", _dig(c3e, ("uniform", "p50_request_ms")))
    put("c3_single_entry_zipf_Glps", _scale(_dig(c3e, ("zipf", "lookups_per_s")), G))
    for tr_, tag_ in (("peer_store", "store"), ("staged_copy", "copy")):
        bt_ = _dig(c3e, ("by_transport", tr_, "uniform")) or {}
        put(f"c3_single_entry_{tag_}_Glps", _scale(bt_.get("lookups_per_s"), G))
        put(f"c3_single_entry_{tag_}_rows_GBps_into_entry", bt_.get("rows_GBps_into_entry_gpu"))
        put(f"c3_single_entry_{tag_}_all_instances_Glps", _scale(_dig(bt_, ("all_instances_at_once", "lookups_per_s")), G))
    put("c3_single_entry_parity", c3e.get("parity"))
    if res.get("multi_gpu_selftest") and c3e.get("by_transport"):
        # which "shard_transport" this run says to ship (tools/choose_transport.py = INTEGRATION.md 4.1; null on logical shards)
        try:
            sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
            import choose_transport
            ch_ = choose_transport.decide(res)
            put("c3_transport_choice", f"{ch_['shard_transport']} ({ch_['confidence']})" if ch_["ok"] else "none: " + (ch_["reasons"][0][:40] if ch_["reasons"] else "?"))
        except Exception as e_:  # noqa: BLE001 — a diagnostic must not cost the line
            put("c3_transport_choice", "error: " + repr(e_)[:40])
    put("c3_single_entry_error", (str(c3e["error"])[:120] if c3e.get("error") else None))
    put("c3_triton_Glps", _scale(g(ex, "c3_sharded_triton", "lookups_per_s"), G))
    put("c3_triton_p50_ms", g(ex, "c3_sharded_triton", "p50_request_ms"))
    put("c3_triton_rows_wrong", g(ex, "c3_sharded_triton", "rows_wrong"))
    put("c3_triton_error", (str(g(ex, "c3_sharded_triton", "error"))[:100] if g(ex, "c3_sharded_triton", "error") else None))
    c3r = ex.get("sharded_c3") or {}
    put("c3_rccl_Glps", _scale(c3r.get("lookups_per_s"), G))
    put("c3_rccl_ranks", c3r.get("ranks"))
    put("c3_rccl_rows_frac_of_link", c3r.get("rows_frac_of_153_GBps_link") or _dig(c3r, ("rccl_groups", "rows_frac_of_153_GBps_link")))
    put("c3_rccl_parity", c3r.get("parity") if "parity" in c3r else c3r.get("parity_vs_oracle_bit_exact"))
    put("c3_rccl_error", (str(c3r["error"])[:120] if c3r.get("error") else None))
    c4 = g(ex, "c4_two_models_triton", "results") or {}
    for k, v in c4.items():   # target_hit_0.5 / 0.9 / 0.99
        if isinstance(v, dict):
            put("c4@" + k.replace("target_hit_", "") + "_Mlps", _scale(v.get("lookups_per_s"), M_))
            put("c4@" + k.replace("target_hit_", "") + "_p50_ms", v.get("p50_request_ms"))
    b8 = g(ex, "c4_two_models_triton", "batched8_target_hit_0.9") or {}
    put("c4_batched8_Mlps", _scale(b8.get("lookups_per_s"), M_))
    put("c4_batched8_p50_execute_ms", b8.get("p50_execute_ms"))
    put("c4_batched8_one_lookup_per_execute", (b8.get("instances_whose_last_execute_was_one_lookup") == 2) if b8 else None)
    c5f = g(ex, "device_driven_tier", "c5_fused_lookup_interact") or {}
    for k, v in c5f.items():   # hit_95 / all_hit
        if isinstance(v, dict):
            put(f"c5_{k}_fused_ms", v.get("fused_ms_per_step"))
            put(f"c5_{k}_separate_ms", v.get("separate_ms_per_step"))
            put(f"c5_{k}_always_fused_ms", v.get("fused_always_ms_per_step"))
    put("c5_dense_frac_of_hbm_peak", g(ex, "c5_lookup_plus_dense", "dense_frac_of_hbm_peak"))
    put("all_hit_2s_host_keys_Glps", _scale(g(ex, "all_hit_two_sessions_host_keys", "lookups_per_s"), G))
    put("all_hit_call_over_kernel_time", rf.get("all_hit_call_over_kernel_time"))
    for tag in ("hit_999", "hit_99", "hit_90", "hit_50"):
        put(f"{tag}_2s_host_keys_Glps", _scale(g(ex, f"{tag}_two_sessions_host_keys", "lookups_per_s"), G))
    put("hit_90_max_call_ms", g(ex, "hit_90_two_sessions_host_keys", "max_call_ms"))
    put("triton_abi_Glps", _scale(g(ex, "triton_abi", "lookups_per_s"), G))
    put("triton_abi_p50_ms", g(ex, "triton_abi", "p50_request_ms"))
    put("triton_abi_p99_ms", g(ex, "triton_abi", "p99_request_ms"))
    put("triton_abi_rows_wrong", g(ex, "triton_abi", "rows_wrong"))
    sl = g(ex, "triton_abi", "slow_requests_ms")
    if sl is not None:
        put("triton_abi_slow_requests", len(sl))
    put("triton_abi_pinned_keys_8B_Glps", _scale(g(ex, "triton_abi", "pinned_keys", "lookups_per_s"), G))
    put("c1_triton_Mlps", _scale(g(ex, "c1_cpu_ps_triton", "lookups_per_s"), M_))
    put("c1_triton_p50_us", _scale(g(ex, "c1_cpu_ps_triton", "p50_request_ms"), 1e3))
    put("wide_keys_95_8B_Glps", _scale(g(ex, "wide_keys_95", "lookups_per_s"), G))
    put("wide_keys_95_frame_of_ref_Glps", _scale(g(ex, "wide_keys_95", "frame_of_reference_default", "lookups_per_s"), G))
    put("device_driven_tier_Glps", _scale(g(ex, "device_driven_tier", "lookups_per_s"), G))
    put("one_session_p50_ms", g(ex, "one_session_host_keys_95", "p50_call_ms"))
    put("device_keys_Glps", _scale(g(ex, "device_keys", "lookups_per_s"), G))
    put("policy_0.9_Glps", _scale(g(ex, "policy_threshold_0.9", "lookups_per_s"), G))
    put("cpu_ps_tier_Mlps", _scale(g(ex, "cpu_parameter_server_tier", "lookups_per_s"), M_))
    c3 = ex.get("sharded_c3_logical") or {}
    put("c3_logical_P", c3.get("shards"))
    put("c3_logical_Glps", _scale(c3.get("lookups_per_s"), G))
    put("c3_logical_parity", c3.get("parity"))
    put("c3_error", (str(c3["error"])[:120] if c3.get("error") else None))
    put("legs_error", (str(ex["legs_error"])[:120] if ex.get("legs_error") else None))
    out["legs"] = legs
    if res.get("per_gpu"):
        out["per_gpu_hit_rate"] = [p.get("measured_hit_rate") for p in res["per_gpu"]]
        out["per_gpu_frac"] = [p.get("frac_of_hbm_peak_1032B_per_lookup") for p in res["per_gpu"]]
    out["checker_note"] = (str(res["checker_note"])[:120] if res.get("checker_note") else None)
    out["extra"] = "bench_extra.json"
    out = _sig(out)
    # the contract's value / ms_per_step keep their full precision
    out["value"], out["ms_per_step"] = res.get("value"), res.get("ms_per_step")
    for drop in ("per_gpu_frac", "per_gpu_hit_rate", "legs", "roofline_pcie"):
        if len(json.dumps(out)) <= limit:
            break
        if drop == "legs":      # keep what fits of the legs, first come first kept
            kept = {}
            for k, v in out["legs"].items():
                kept[k] = v
                out["legs"] = kept
                if len(json.dumps(out)) > limit:
                    del kept[k]
                    break
        else:
            out.pop(drop, None)
    return out


def _dig(d, ks):
    for k in ks:
        if not isinstance(d, dict):
            return None
        d = d.get(k)
    return d


def _scale(v, f):
    return None if v is None else v * f


def emit(res):
    """Full result -> bench_extra.json (cwd and, when it exists, gpurun_out/); compact line -> stdout, LAST."""
    text = json.dumps(res)
    for d in (Path.cwd(), ROOT / "gpurun_out"):
        try:
            if d.is_dir():
                (d / "bench_extra.json").write_text(text + "\n")
        except OSError as e:
            sys.stderr.write(f"[bench] bench_extra.json not written in {d}: {e!r}\n")
    sys.stdout.flush()
    try:
        # C stdio of the libraries in this process (RCCL prints a version banner to stdout through it, buffered until exit when
        # stdout is a pipe): out now, so that the line below stays the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    print(json.dumps(compact_line(res)), flush=True)


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # N GPUs = the reference's arrangement (docs/architecture.md:11,29; hps_backend/src/model_state.cpp:395-420;
    # backend.cpp:68-69): ONE process with ONE parameter server (one copy of the host tier) and one embedding cache per
    # GPU of deployed_device_list, the lookup sessions of every GPU driven by host threads of that process.  Under
    # torch.distributed.run rank 0 is that process; the other ranks take no part in the replicas measurement (they wait
    # at a gloo barrier) and come in only for the sharded-table leg, where each rank owns one GPU (RCCL inside the engine).
    n_rep = max(world, a.gpus, 1)
    if a.direct < 0:
        # the host-gather tier needs host cores (its gather runs on ~13 threads per GPU) and host DRAM bandwidth (every missed
        # row is read by a host thread, written to the staging buffer and read again by the DMA engine: ~3 x 47 GB/s per GPU at
        # this workload's PCIe-bound rate — beyond four GPUs that is more than a two-socket host delivers); the device-driven
        # tier needs no host thread and reads every missed row once
        a.direct = 0 if (effective_cpus() >= 12 * n_rep and n_rep <= 4) else 1

    # HIP spreads a process's streams over 4 hardware queues by default; two lookup sessions whose streams land on the
    # same queue run strictly one after the other.  Must be in the environment before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if n_rep > 1 and not a.direct:
        # host-gather tier on N GPUs: the serving pool of the one process gathers for all of them (default: at most 64 threads)
        os.environ.setdefault("HPS_SERVING_THREADS", str(max(1, min(effective_cpus() - 3, 64 * n_rep))))
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        # CPU rendezvous only: nothing of the data path goes through torch.distributed
        dist.init_process_group("gloo", timeout=datetime.timedelta(hours=6))
    if rank != 0:
        idle_rank(a, dist, rank, world, local_rank)
        return
    # torch sizes its CPU thread pool (OpenMP) by the machine's hardware threads — 256 on the GPU boxes, whose containers have a
    # 16-CPU quota.  A parallel copy (tensor.cpu(), pin_memory()) then leaves 255 workers busy-waiting, the cgroup runs out of
    # quota and EVERY thread of the process is frozen until the 100-ms period ends: round 2's "87-ms call on page-locked keys"
    # (always that leg: it starts right after 28 pin_memory() copies; profiles/round3/legs_three_runs.txt shows 323 ms of
    # throttled time in exactly the leg with the 86.7-ms call).  torch is this harness's plumbing, not the product: keep it
    # inside the quota.
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, effective_cpus() // 2))))
    from hugectr_backend_amd.gpu_wait import bind_process_to_numa_node, gpu_numa_nodes, wait_for_gpu
    wait_for_gpu(30.0)   # a device that another process has just released can be invisible for a moment
    # Two-socket host: this process (its session threads, the HIP runtime's threads, the request buffers it allocates) stays on the
    # socket its GPU hangs off, as a deployment does with `numactl --cpunodebind`: unbound, the same binary gave 1.50-2.09 G lookups/s
    # from run to run on one box (key staging 0.19-0.50 ms and host gather 0.43-0.85 ms per call, wherever threads and pages had
    # landed), under `taskset` to ONE node 2.03-2.06 G in every run (profiles/round5/numa_one_node.txt).  Only when every GPU of
    # the run hangs off the same node; --numa-bind 0 leaves the process alone.  Before HIP starts: its threads inherit the mask.
    bound_node = -1
    if a.numa_bind and os.path.exists("/sys/devices/system/node/node1"):
        nodes = gpu_numa_nodes()
        used = {nodes[g % len(nodes)] for g in range(n_rep)} if nodes else set()
        if len(used) == 1 and min(used) >= 0 and bind_process_to_numa_node(min(used)):
            bound_node = min(used)
    import torch
    torch.set_num_threads(max(1, min(8, effective_cpus() // 2)))
    from hugectr_backend_amd import build as hb
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs an MI355X"
    # replica g runs on device g.  Only when fewer GPUs than replicas are visible (the 1-GPU development box) do replicas
    # share devices — each extra replica of a device then deploys the model a second time under its own name (own tables,
    # own cache), which exercises the same multi-cache code path.
    devs = [g % ndev for g in range(n_rep)]
    shared_gpu = n_rep > ndev
    dev = devs[0]
    torch.cuda.set_device(dev)
    numa_note = None
    hb.build()
    from hugectr_backend_amd import hps

    # ---- N > 1: first contact with the machine's links, BEFORE anything is trusted with them (hps_multi_gpu_selftest): peer access
    # matrix, one 4-KB peer store + read-back per ordered pair, GB/s per pair of kernel stores over the peer mapping and of
    # hipMemcpyPeerAsync (the two transports of the table-sharded lookup), one RCCL all-reduce of one word with one rank per GPU —
    # behind a deadline, so that a hang is ATTRIBUTED (the report names the step) before the legs start.  What it finds decides
    # which config-3 legs run: no peer stores -> staged_copy only; RCCL stuck -> no RCCL leg.
    selftest = None
    c3_transports = ("peer_store", "staged_copy")
    rccl_usable = True
    if n_rep > 1 and not a.no_selftest:
        t_st = time.time()
        try:
            selftest = hps.multi_gpu_selftest(sorted(set(devs)), 64 << 20, a.selftest_timeout, with_rccl=not a.no_c3_leg)
        except Exception as e:  # noqa: BLE001
            selftest = {"error": repr(e)[:200]}
        selftest["wall_seconds"] = time.time() - t_st
        stuck = str(selftest.get("stuck_in") or "")
        flat = lambda m: [v for row in (m or []) for v in row if v is not None]   # noqa: E731
        stores_ok = not selftest.get("timeout") and not selftest.get("error") and all(v == 1 for v in flat(selftest.get("peer_access"))) \
            and all(v == 1 for v in flat(selftest.get("store_4k_ok")))
        if not stores_ok and (len(set(devs)) > 1):
            c3_transports = ("staged_copy",)
        if stuck.startswith("RCCL") or (selftest.get("rccl_allreduce") or {}).get("ok") is False:
            rccl_usable = False
        if selftest.get("timeout") and not stuck.startswith("RCCL"):
            # a peer store or a peer copy did not come back: the device it ran on is in an unknown state — report and stop here
            sys.stderr.write(f"[bench] multi-GPU self-test stuck in: {stuck}\n")
        sys.stderr.write(f"[bench] multi_gpu_selftest: {json.dumps(selftest)[:1500]}\n")

    T, R, D, B = a.tables, a.rows, a.dim, a.batch
    N = T * B
    K, W = a.steps, a.warmup
    blocks = a.blocks if a.blocks > 0 else int(min(12, max(1, -(-240 // max(K, 1)))))
    nb = W + K * blocks
    if a.distinct_batches > 0:
        nb = min(nb, a.distinct_batches)
    # one model per replica that has to share a device with an earlier one (development box only); on an N-GPU node
    # there is ONE model deployed on N devices
    models, rep_model, _ = plan_replicas(n_rep, ndev)
    model = models[0]
    # host-memory guard: the host tier exists ONCE whatever the number of GPUs (one parameter server per process); what
    # grows with N is the timed region's key batches.  Only if the box cannot hold that do rows/table shrink, and the
    # workload string says so.
    rows_requested = R
    budget = int(host_memory_budget() * 0.85) - (12 << 30) - n_rep * nb * N * 8
    per_row = len(models) * T * (4 * D + 64) + T * 56 + 2 * (4 * D + 48)   # tables + index, the oracle's index of every table, its copy of two
    if R * per_row > budget:
        R = max(B, int(budget // per_row))
    setup_note = ""
    threshold = 1.0 if a.mode == "sync" else 0.5
    cfg = {
        "supportlonglong": True,
        "volatile_db": {"type": "hash_map", "num_partitions": 8},
        "models": [{
            "model": m,
            "sparse_files": [f"synthetic://t{t}" for t in range(T)],
            "num_of_worker_buffer_in_pool": max(3, a.sessions),
            "embedding_vecsize_per_table": [D] * T,
            "maxnum_catfeature_query_per_table_per_sample": [1] * T,
            "default_value_for_each_table": [0.0] * T,
            "deployed_device_list": plan_replicas(n_rep, ndev)[2][m],
            "max_batch_size": B,
            "gpucache": True,
            "gpucacheper": a.cache_frac,
            "hit_rate_threshold": threshold,
            "ps_direct_access": bool(a.direct),
        } for m in models],
    }
    if os.environ.get("BENCH_PS_EXTRA"):          # harness-only: extra ps.json model keys for A/B runs (tools/ab_small_insert.sh)
        for mm in cfg["models"]:
            mm.update(json.loads(os.environ["BENCH_PS_EXTRA"]))

    def setup():
        """One parameter server: the host tables are generated once, then one cache per (model, device)."""
        t_setup = time.time()
        ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
        for m in models:
            for t in range(T):
                ps.load_table_synthetic(m, t, SEED, 0, R)
        t_tables = time.time() - t_setup
        for m in models:
            ps.create_embedding_cache_per_model(m)
        caches = [ps.get_embedding_cache(rep_model[g], devs[g]) for g in range(n_rep)]
        t_cache = time.time() - t_setup - t_tables
        return ps, caches, t_tables, t_cache

    # ps_direct_access pins the whole host tier (hipHostMalloc).  Where the box refuses that much page-locked
    # memory, the run falls back to the host-gather tier of the same library (never to a CPU path) and the
    # result line says so in config.ps_tier.
    direct_note = None
    made = None
    if a.direct:
        try:
            made = setup()
        except Exception as e:  # noqa: BLE001
            direct_note = f"ps_direct_access unavailable on this box ({str(e)[:160]})"
            sys.stderr.write(f"[bench] {direct_note}; falling back to the host-gather tier\n")
            import gc
            gc.collect()
            a.direct = 0
            for mc in cfg["models"]:
                mc["ps_direct_access"] = False
    if made is None:
        made = setup()
    ps, caches, t_tables, t_cache = made
    cache = caches[0]
    split = (a.split_probe != 0) and not a.direct     # host-gather tier only
    C = int(np.ceil(a.cache_frac * R))
    cdf_h = torch.from_numpy(zipf_cdf(C, a.zipf))

    class Replica:
        """One GPU's share of the deployment: its cache, its lookup sessions, its output buffers, its key batches."""

    reps = []
    for g in range(n_rep):
        rp = Replica()
        rp.g, rp.dev, rp.model, rp.cache = g, devs[g], rep_model[g], caches[g]
        with torch.cuda.device(rp.dev):
            rp.sessions = [hps.LookupSession.create(ps, rp.model, rp.cache) for _ in range(a.sessions)]
            for s in rp.sessions:
                s.set_option("timing", 1)
                s.set_option("probe_variant", a.probe_variant)
                s.set_option("xcd_walk", a.xcd_walk)
                s.set_option("split_probe", 1 if split else 0)
                s.set_option("narrow_keys", a.narrow_keys)
                s.set_option("chain_gather", a.chain_gather)
                s.set_option("probe_in_lane", a.probe_in_lane)
                s.set_option("keys_by_kernel", a.keys_by_kernel)
            # resident set = what the warm-up actually placed (first C rows in file order minus over-full buckets)
            resident = []
            for t in range(T):
                k = np.arange(C, dtype=np.int64)
                resident.append(k[rp.cache.query(t, k) >= 0])
            rp.resident_frac = float(np.mean([r.size / C for r in resident]))
            # ---- timed region layout: W warm-up steps, then `blocks` blocks of exactly K steps, every step a fresh batch ----
            gen = torch.Generator(device=f"cuda:{rp.dev}")
            gen.manual_seed(SEED + g)
            rp.cdf_d = cdf_h.cuda()
            rp.resident_d = [torch.from_numpy(r).cuda() for r in resident]
            rp.run = Runner(torch, hps, rp.sessions, T, B, D, rp.dev)
            # The reference's contract (docs/architecture.md:308-323; hps.cc:586-597): the keys of a request are in HOST
            # memory.  Generated on the device (fast), then moved to ordinary pageable numpy arrays, one per batch.
            rp.host_batches = []
            for i in range(0, nb, 16):
                for bt_ in make_batches_gpu(torch, gen, rp.resident_d, rp.cdf_d, R, C, B, a.hit, min(16, nb - i)):
                    arr = bt_.cpu().numpy()
                    rp.host_batches.append((arr, rp.run.pack_host(arr)))
            torch.cuda.synchronize()
            if g > 0:
                del rp.resident_d, rp.cdf_d
        rp.rec = []
        reps.append(rp)
    sessions, run, host_batches = reps[0].sessions, reps[0].run, reps[0].host_batches
    resident_d, cdf_d, resident_frac = reps[0].resident_d, reps[0].cdf_d, reps[0].resident_frac
    ncpu = effective_cpus()

    def barrier():
        # every replica lives in this process: the barrier between blocks is the join of their threads plus a
        # synchronisation of every device (no other process takes part in the measurement)
        for d in sorted(set(devs)):
            torch.cuda.synchronize(d)

    def run_all(count, first, record):
        """`count` steps on EVERY replica, concurrently (each replica's sessions share its steps); returns when all are done."""
        if n_rep == 1:
            reps[0].run.run(reps[0].host_batches, count, first, "host", record=reps[0].rec if record else None)
            return
        th = [threading.Thread(target=rp.run.run, args=(rp.host_batches, count, first, "host"),
                               kwargs={"record": rp.rec if record else None}) for rp in reps]
        [x.start() for x in th]
        [x.join() for x in th]

    # no collector pass of the interpreter inside the timed region (a full pass over a torch process's objects takes tens of
    # milliseconds and holds the GIL the session threads need between two calls)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    run_all(W, 0, False)
    if a.mode == "async":
        for c in caches:
            c.wait_async()
    block_s = []
    thr0 = cpu_throttle_stat()
    ct0, wall0 = cpu_times(), time.perf_counter()
    for blk in range(blocks):
        barrier()
        t0 = time.perf_counter()
        run_all(K, W + blk * K, True)
        barrier()
        block_s.append(time.perf_counter() - t0)   # until the LAST replica has finished its K steps (= max over GPUs)
    thr1 = cpu_throttle_stat()
    ct1, wall1 = cpu_times(), time.perf_counter()
    gc.enable()
    elapsed = float(np.median(block_s))
    main_rec = [r for rp in reps for r in rp.rec]
    per_gpu = None
    if n_rep > 1:
        per_gpu = []
        for rp in reps:
            m_ = summarize(rp.rec, N, D, elapsed, K)
            per_gpu.append({"replica": rp.g, "device": rp.dev, "model": rp.model,
                            "p50_call_ms": m_["p50_call_ms"], "p99_call_ms": m_["p99_call_ms"],
                            "probe_ms": m_["probe_ms"], "gather_ms": m_["gather_ms"], "scatter_ms": m_["scatter_ms"],
                            "insert_ms": m_["insert_ms"], "measured_hit_rate": m_["measured_hit_rate"],
                            "frac_of_hbm_peak_1032B_per_lookup": m_["frac_of_hbm_peak_1032B_per_lookup"],
                            "resident_fraction_after_warmup": rp.resident_frac})
        # every other replica's answer to one whole batch against the rows as they sit in the host tier (keys are 0..R-1 in
        # file order: the expected row of key k is row k); replica 0 gets the full set of checks further down
        if a.mode == "sync":
            views = [ps.table_data(rp_.model, t) for rp_ in reps[:1] for t in range(T)] if len(models) == 1 else None
            for rp in reps[1:]:
                q = rp.host_batches[0][0]
                rp.sessions[0].lookup_packed(rp.host_batches[0][1], rp.run.vptrs[0], rp.run.counts)
                torch.cuda.synchronize(rp.dev)
                got = rp.run.outs[0].cpu().numpy().reshape(N, D)
                okr = True
                for t in range(T):
                    tk, tr = views[t] if views else ps.table_data(rp.model, t)
                    qt = q[t * B:(t + 1) * B]
                    okr &= bool(np.array_equal(tr[qt].view(np.uint32), got[t * B:(t + 1) * B].view(np.uint32)))
                per_gpu[rp.g]["parity_full_batch_vs_direct_row_index"] = okr
                del got
        # the other replicas' sessions and batches are released before the checks (their caches go with the server)
        for rp in reps[1:]:
            for s in rp.sessions:
                s.close()
            rp.sessions, rp.host_batches, rp.run = [], [], None

    # what THIS box's HBM delivers to a plain device-to-device copy (read + write bytes over time): the boxes of the pool differ by
    # up to 17 % in their HBM-bound kernels with identical code and traffic, and the line should say which kind this one is.
    # (After the timed region: allocating and freeing the 2 GiB right before it made the first timed block 40 % slower.)
    box_copy_gbs = None
    try:
        with torch.cuda.device(dev):
            xa = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
            xb = torch.empty_like(xa)
            xa.fill_(1.0)
            for _ in range(3):
                xb.copy_(xa)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                xb.copy_(xa)
            e1.record()
            torch.cuda.synchronize()
            box_copy_gbs = 10 * 2 * xa.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del xa, xb
            torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        sys.stderr.write(f"[bench] copy-bandwidth probe skipped: {e!r}\n")

    # ---- extra legs (outside the timed region; per-GPU numbers of this rank) --------------------------------
    extra = {}
    if not a.no_extra_legs and n_rep == 1:
        gen2 = torch.Generator(device="cuda")
        gen2.manual_seed(SEED + 1000 + rank)

        def fresh(n, hit=None, res=None):
            return make_batches_gpu(torch, gen2, resident_d if res is None else res, cdf_d, R, C, B, a.hit if hit is None else hit, n)

        def resident_now():
            """The keys of the warm range that sit in the cache NOW (the timed region and the legs before have evicted some)."""
            out = []
            k = np.arange(C, dtype=np.int64)
            for t in range(T):
                out.append(torch.from_numpy(k[cache.query(t, k) >= 0]).cuda())
            return out

        def host_form(batches):
            hb2 = [(x.cpu().numpy(),) for x in batches]
            return [(x[0], run.pack_host(x[0])) for x in hb2]

        def leg(batches, steps, mode, sess=None):
            r = []
            # as in the timed region: no collector pass of the interpreter inside a leg.  A pass that finds dead tensors frees
            # device and page-locked memory (hipFree / hipHostFree: both wait for the device and hold the runtime's lock) —
            # round 2's 87-ms calls "on page-locked keys" were such passes landing in whichever leg was running
            gc.collect()
            gc.disable()
            run.run(batches, 4, 0, mode, sess)
            torch.cuda.synchronize()
            th0, st0 = cpu_throttle_stat(), cpu_times()[0]
            tl0 = time.perf_counter()
            run.run(batches, steps, 4, mode, sess, record=r)
            torch.cuda.synchronize()
            out = summarize(r, N, D, time.perf_counter() - tl0, steps)
            gc.enable()
            th1, st1 = cpu_throttle_stat(), cpu_times()[0]
            if st0 is not None and st1 is not None:
                out["hypervisor_steal_ms"] = (st1 - st0) * 1e3
            # the slowest call of the leg, split as the engine saw it: [end to end (Python clock), until the miss counts are on
            # the host, host gather, upload tail + scatter + insert, whole call inside the engine, key staging]
            sl = max(r, key=lambda x: x[0])
            out["slowest_call_ms"] = [round(sl[0], 3)] + [round(float(x), 3) for x in sl[8]] + [round(sl[9], 3)]
            if th0 and th1:
                # the container's CPU quota (cgroup cpu.max): a throttled period freezes every thread of the process until the
                # period ends — calls of tens of milliseconds that have nothing to do with the path
                out["cpu_quota_throttled_periods"] = th1[0] - th0[0]
                out["cpu_quota_throttled_ms"] = (th1[1] - th0[1]) / 1e3
            return out

        def run_legs():
            # (1) KEYS already in HBM (hps_session_lookup_device, an addition to the reference's API): what the path does when
            #     nothing but missed rows crosses PCIe — round 1's headline
            dk = fresh(28)
            extra["device_keys"] = leg(dk, 24, "device")
            extra["device_keys"]["note"] = "hps_session_lookup_device: KEYS resident in HBM, same cache and sessions"
            # (2) host keys in page-locked memory, one flat array per request (what Triton's pinned input pool hands over)
            pinned = []
            for bt_ in dk:
                p = bt_.cpu().pin_memory()
                pinned.append((p, run.pack_host(p.numpy())))
            extra["pinned_host_keys"] = leg(pinned, 24, "pinned")
            extra["pinned_host_keys"]["note"] = ("the device_keys leg's batches again (their misses are resident by now: see measured_hit_rate) — "
                                                 "this leg and the two after it compare key transports, not miss paths")
            for s in sessions:
                s.set_option("narrow_keys", 0)
            extra["pinned_host_keys_8_byte_dma_in_place"] = leg(pinned, 24, "pinned")
            hb_ = [(x.cpu().numpy(),) for x in dk]
            hb_ = [(x[0], run.pack_host(x[0])) for x in hb_]
            extra["pageable_host_keys_8_byte_staging"] = leg(hb_, 24, "host")
            for s in sessions:
                s.set_option("narrow_keys", a.narrow_keys)
            del pinned, hb_
            # (2b) ONE session, fresh batches at the headline's hit rate: the latency of a request that has the GPU and the link
            #      to itself (the headline's p50 is that of two sessions sharing both)
            one = [(x.cpu().numpy(),) for x in fresh(28)]
            one = [(x[0], run.pack_host(x[0])) for x in one]
            extra["one_session_host_keys_95"] = leg(one, 24, "host", [0])
            extra["one_session_host_keys_95"]["note"] = "same workload, a single lookup session: request latency without a second session on the GPU and the link"
            del one
            # (2c) the headline's two sessions at a MEASURED hit rate of 0.950 +- 0.001 (the headline itself measures ~0.957: "at
            #      least 95 %" — this leg is the rate at exactly 95.0 %).  The draw probability is corrected by what a first pass measured.
            h_try, r_ = 0.950, None
            for attempt in range(3):
                nb_ = fresh(28, h_try, resident_now())
                r_ = leg(host_form(nb_), 24, "host")
                del nb_
                got_ = r_["measured_hit_rate"]
                if abs(got_ - 0.950) <= 0.001:
                    break
                h_try = min(1.0, max(0.0, h_try + 0.950 - got_))
            extra["hit_950_two_sessions_host_keys"] = dict(r_, resident_draw_probability=h_try, passes=attempt + 1)
            # (2d) SERVING UNDER A CACHE REFRESH (the reference runs refresh_embedding_cache on a timer next to the lookups:
            #      model_state.cpp:125-178, 413-427).  A thread loops FULL refreshes of this cache — every resident row re-read from the
            #      host tier and uploaded, the reference's behaviour, 26.6 GB here — while the two sessions serve fresh batches at the
            #      headline's hit rate, then at 99.9 %.  (The default refresh takes only rows that can differ: nothing at all for these
            #      unchanged tables — its cost is the unchanged_refresh figure.)
            def under_refresh(batches, steps):
                stop, acc = threading.Event(), {"passes": 0, "row_bytes": 0, "seconds": 0.0}
                def bg():
                    try:
                        while not stop.is_set():
                            st_ = ps.refresh_embedding_cache(model, dev, full=True)
                            acc["passes"] += 1
                            acc["row_bytes"] += st_["row_bytes"]
                            acc["seconds"] += st_["seconds"]
                    except Exception as e_:  # noqa: BLE001
                        acc["error"] = repr(e_)[:200]
                c0 = cache.refresh_rows_uploaded()
                th_ = threading.Thread(target=bg, daemon=True)
                th_.start()
                t_w = time.time()
                while cache.refresh_rows_uploaded() == c0 and time.time() - t_w < 20 and "error" not in acc:
                    time.sleep(0.005)      # (the pass starts with a read-back of the resident keys)
                c1, t1_ = cache.refresh_rows_uploaded(), time.perf_counter()
                r_ = leg(batches, steps, "host")
                c2, t2_ = cache.refresh_rows_uploaded(), time.perf_counter()
                stop.set()
                th_.join(60)
                r_["refresh_rows_during_leg"] = int(c2 - c1)
                r_["refresh_GBps_during_leg"] = (c2 - c1) * 4 * D / max(t2_ - t1_, 1e-9) / 1e9
                r_["refresh_passes_completed"] = acc["passes"]
                r_["refresh_GBps_whole_passes"] = acc["row_bytes"] / acc["seconds"] / 1e9 if acc["seconds"] > 0 else None
                if "error" in acc:
                    r_["refresh_error"] = acc["error"]
                return r_

            t_u = time.perf_counter()
            st_u = ps.refresh_embedding_cache(model, dev)
            extra["unchanged_refresh"] = dict(st_u, wall_ms=(time.perf_counter() - t_u) * 1e3,
                                              note="default refresh of a cache whose tables did not change: nothing is re-read or uploaded")
            extra["headline_under_refresh"] = under_refresh(host_form(fresh(64)), 60)
            extra["hit_999_under_refresh"] = under_refresh(host_form(fresh(64, 0.999, resident_now())), 60)
            # (3) every key resident: the GPU-side ceiling of the path, one session (kernels run alone)
            #     TRUE all-hit: the keys are drawn from what is resident at this moment (round 3 drew them from the set resident
            #     after warm-up; a few hundred of those had been evicted by then and the "all-hit" legs measured a miss path)
            res_now = resident_now()
            hot = fresh(8, 1.1, res_now)
            extra["all_hit_one_session_device_keys"] = leg(hot, 24, "device", [0])
            extra["all_hit_two_sessions_host_keys"] = leg(host_form(hot), 40, "host")
            del hot
            # (3b) near-all-hit, where a production cache lives: 99.9 % and 99 % of the keys resident, fresh cold keys in every batch
            #      (two sessions, host keys; the cold keys are inserted, so the resident set is taken again before each leg)
            for tag, h_ in (("hit_999", 0.999), ("hit_99", 0.99)):
                nb_ = fresh(44, h_, resident_now())
                extra[f"{tag}_two_sessions_host_keys"] = leg(host_form(nb_), 40, "host")
                del nb_
            del res_now
            # (4) the reference's default policy (hit_rate_threshold 0.9): the hit rate over the call's UNIQUE keys decides;
            #     async tables return the default vector for their misses and are filled in the background
            fa = [(x.cpu().numpy(),) for x in fresh(28)]
            fa = [(x[0], run.pack_host(x[0])) for x in fa]
            for s in sessions:
                s.set_option("hit_rate_threshold_permille", 900)
            extra["policy_threshold_0.9"] = leg(fa, 24, "host")
            extra["policy_threshold_0.9"]["calls_answered_async"] = int(sum(s.last_stats().async_insert for s in sessions))
            cache.wait_async()
            for s in sessions:
                s.set_option("hit_rate_threshold_permille", 1000 if a.mode == "sync" else 500)
            del fa
            # (5) BASELINE config 5: the dense step (bottom MLP 13-512-256-D + dot interaction, fp16 MFMA) consuming
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
                    ops[0].forward(xd, run.outs[0], B, out=outd[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    ops[0].forward(xd, run.outs[0], B, out=outd[0])
                e1.record()
                torch.cuda.synchronize()
                dense_ms = e0.elapsed_time(e1) / 20
                dense_bytes = N * 4 * D + B * 13 * 4 + 2 * B * D * 2 + B * ops[0].out_stride * 2
                dense_flops = 2 * B * (16 * 512 + 512 * 256 + 256 * D) + 2 * B * 32 * 32 * D
                run.post_hooks[:] = [lambda si: (ops[si].forward(xd, run.outs[si], B, out=outd[si]),
                                                 torch.cuda.current_stream().synchronize()) for _ in sessions]
                c5 = leg(fresh(28), 24, "device")
                run.post_hooks[:] = []
                c5.update({"dense_kernels_ms": dense_ms, "dense_algorithmic_bytes": dense_bytes,
                           "dense_frac_of_hbm_peak": dense_bytes / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "dense_mfma_tflops": dense_flops / (dense_ms * 1e-3) / 1e12,
                           "samples_per_s": c5["lookups_per_s"] / T,
                           "note": "lookup (device keys, sync insert, exact rows) + bottom MLP 13-512-256-%d + dot interaction "
                                   "per step; output [batch, %d] f16" % (D, ops[0].out_dim)})
                extra["c5_lookup_plus_dense"] = c5
                if a.direct and a.mode == "sync":
                    run.step_hooks[:] = [lambda si, keys: ops[si].lookup_interact(sessions[si], keys, B, xd, out=outd[si]) for _ in sessions]
                    c5f = leg(fresh(28), 24, "device")
                    run.step_hooks[:] = []
                    c5f.update({"samples_per_s": c5f["lookups_per_s"] / T,
                                "note": "one call per step: probe, miss fetch, bottom MLP, interaction reading cache slots / staging, insert"})
                    extra["c5_fused_lookup_interact"] = c5f
            # (6) the build's own CPU parameter server (the host tier, `gpucache=false` path of the reference:
            #     docs/architecture.md:72) on the identical batches: hps_server_fetch per table, all host cores
            outc = np.empty((B, D), dtype=np.float32)   # one table's slice, reused (a fresh 33-MB array per fetch is page faults)
            tcp0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - tcp0 < 4.0 and reps < 40:
                q = host_batches[reps % len(host_batches)][0]
                for t in range(T):
                    qt = q[t * B:(t + 1) * B]
                    hps._check(hps.LIB.hps_server_fetch(ps._h, model.encode(), t, qt.ctypes.data, B, outc.ctypes.data, None))
                reps += 1
            tcp = time.perf_counter() - tcp0
            extra["cpu_parameter_server_tier"] = {
                "lookups_per_s": reps * N / tcp, "cores": ncpu,
                "note": f"hps_server_fetch (host tier of this build) on the timed region's own batches, all {T} tables, "
                        f"{reps} passes, rows into pageable host memory"}
            del outc
            # (7) the rest of SURVEY 8(d)'s hit-rate sweep (h = 1.0 / 0.999 / 0.99 are legs (3) / (3b), 0.957 is the headline): 90 % and
            #     50 % of the keys resident — 87 MB / 436 MB of missed rows per call over PCIe.  Last, because these calls replace most
            #     of what the cache holds
            for tag, h_, st_ in (("hit_90", 0.90, 12), ("hit_50", 0.50, 8)):
                nb_ = fresh(st_ + 4, h_, resident_now())
                extra[f"{tag}_two_sessions_host_keys"] = leg(host_form(nb_), st_, "host")
                del nb_

        try:   # the legs are informational: a failure in one of them must not cost the headline line
            run_legs()
        except Exception as e:  # noqa: BLE001
            extra["legs_error"] = repr(e)[:300]
            sys.stderr.write(f"[bench] extra legs stopped: {e!r}\n")
        finally:
            run.post_hooks[:] = []
            run.step_hooks[:] = []
            for s in sessions:
                s.set_option("narrow_keys", a.narrow_keys)
                s.set_option("hit_rate_threshold_permille", 1000 if a.mode == "sync" else 500)
    del cdf_d, resident_d
    reps[0].cdf_d = reps[0].resident_d = None

    # ---- untimed checks of one step against the CPU oracle, and the cpu_baseline leg ----
    parity = parity_full = None
    cpu = None
    checker_note = None
    if rank == 0:
        def run_checker():
            nonlocal parity, parity_full, cpu
            from oracle import hps_oracle as O
            q = host_batches[0][0]
            sessions[0].lookup_packed(host_batches[0][1], run.vptrs[0], run.counts)
            torch.cuda.synchronize()
            got_all = run.outs[0].cpu().numpy().reshape(N, D)
            # (a) the oracle with ITS OWN copy of the first two tables (rows from the oracle's restatement of the recipe)
            chk_tables = min(2, T)
            co = O.COracle()
            keys_seq = np.arange(R, dtype=np.int64)
            for t in range(chk_tables):
                rows = np.empty((R, D), dtype=np.float32)
                step = (R + ncpu - 1) // ncpu

                def gen_(lo, t=t, rows=rows):
                    hi = min(R, lo + step)
                    if hi > lo:
                        O.c_synth_rows(SEED, t, lo, hi - lo, D, out=rows[lo:hi])

                th = [threading.Thread(target=gen_, args=(lo,)) for lo in range(0, R, step)]
                [x.start() for x in th]
                [x.join() for x in th]
                co.add_table_arrays(keys_seq, rows)
            ref = co.lookup(q[: chk_tables * B], [B] * chk_tables, [0.0] * chk_tables, threads=ncpu).reshape(-1, D)
            got = got_all[: chk_tables * B]
            if a.mode == "sync":
                parity = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
            else:   # async mode: resident keys exact, others default
                same = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=1)
                parity = bool((same | (got == 0.0).all(axis=1)).all())
            co.close()
            del co
            # (b) the WHOLE batch, all tables, with no hashing on the checking side: the tables' keys are 0..R-1 in file
            #     order, so the expected row of key k is row k of the table as it sits in the host tier
            if a.mode == "sync":
                okf = True
                for t in range(T):
                    tk, tr = ps.table_data(model, t)
                    qt = q[t * B:(t + 1) * B]
                    okf &= bool(np.array_equal(tk[qt], qt))
                    okf &= bool(np.array_equal(tr[qt].view(np.uint32), got_all[t * B:(t + 1) * B].view(np.uint32)))
                parity_full = okf
            if a.no_cpu_baseline:
                return
            # ---- cpu_baseline: the oracle (a port of the reference's hash_map parameter-server lookup: a hash-map find per
            # key, row copy or default) on the timed region's own batches, ALL tables, all host cores.  The oracle builds its
            # own index over every table; the rows it copies are the host tier's (borrowed, read-only: no second 133 GB). ----
            cb = O.COracle()
            views = [ps.table_data(model, t) for t in range(T)]
            errs = []

            def add(t):
                try:
                    cb.add_table_arrays_borrowed(t, *views[t])
                except Exception as e:  # noqa: BLE001
                    errs.append(e)

            cb.reserve(T)
            th = [threading.Thread(target=add, args=(t,)) for t in range(T)]
            for i in range(0, T, ncpu):
                [x.start() for x in th[i:i + ncpu]]
                [x.join() for x in th[i:i + ncpu]]
            if errs:
                raise errs[0]
            outc = np.empty(N * D, dtype=np.float32)
            done, reps, tc0 = 0, 0, time.perf_counter()
            while time.perf_counter() - tc0 < a.cpu_seconds and reps < 2000:
                qb = host_batches[reps % len(host_batches)][0]
                cb.lookup(qb, [B] * T, [0.0] * T, threads=ncpu, out=outc)
                done += qb.size
                reps += 1
            tcpu = time.perf_counter() - tc0
            cpu = {"value": done / tcpu, "unit": "lookups/s", "cores": ncpu, "kind": "port",
                   "sample": f"{reps} whole batches of the timed region ({N} keys each, all {T} tables of {R} rows x {D} fp32), "
                             f"oracle/hps_oracle.c oracle_lookup_mt with {ncpu} threads, rows written to pageable host memory"}
            cb.close()

        try:   # the oracle is the checker; if it cannot run here the measurement still stands, marked unchecked
            run_checker()
        except Exception as e:  # noqa: BLE001
            checker_note = repr(e)[:300]
            sys.stderr.write(f"[bench] oracle check / cpu baseline stopped: {e!r}\n")

    res = None
    if rank == 0:
        m = summarize(main_rec, N, D, elapsed, K)
        ph = np.array([r[8] for r in main_rec])
        lat = np.array([r[0] for r in main_rec])
        gpu_ms = np.array([r[7] for r in main_rec])
        uniq = float(np.mean([r[6] for r in main_rec]))
        hits = N * m["measured_hit_rate"]
        probe, gather, scatter = m["probe_ms"], m["gather_ms"], m["scatter_ms"]
        hbm_ms = probe + gather + scatter
        alg = N * (8 + 8 * D)                       # 8 B key + 4D row read + 4D row write per lookup
        # the primary figure counts the cache-insert kernel too: since round 4 it is off the call's return path, but it occupies
        # the GPU on every call that missed (and serialises with the other session's probe through the writer event) — the three
        # kernels that produce the call's rows alone are `frac_return_path_kernels`
        achieved_rows = alg / (hbm_ms * 1e-3) / 1e9
        achieved = alg / ((hbm_ms + m["insert_ms"]) * 1e-3) / 1e9
        # HBM traffic of the kernels comes from the committed rocprofv3 PMC passes (bench.py cannot run the profiler on
        # itself): profiles/pmc_latest.json, written by tools/summarize_profile.py
        traffic = traffic_src = None
        try:
            pj = json.loads((ROOT / "profiles" / "pmc_latest.json").read_text())
            if pj.get("workload_keys") == N and pj.get("dim") == D and pj.get("round", 1) >= 2:
                traffic = pj["pmc"]["hbm_bytes_per_call_fetch_doubled"]
                traffic_src = pj.get("source")
        except Exception:
            pass
        fetch_ms = float(ph[:, 1].mean()) if ph.size else 0.0
        res = {
            "metric": "embedding lookups/sec, Criteo 26-slot 64K batch",
            "value": n_rep * K * N / elapsed,
            "unit": "lookups/s",
            "n_gpus": n_rep,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp32 rows / int64 keys (moved, never computed)",
            "data": "synthetic",
            "config": {
                "workload": f"Criteo DLRM {T} sparse slots, {R} rows/table"
                            + (f" (requested {rows_requested}; reduced to fit this box's host memory{setup_note})" if R != rows_requested else "")
                            + f" x {D}-dim, {B} batch ({N} keys) per GPU and step, "
                            f"gpucacheper {a.cache_frac}, >=95% hit (draw {a.hit}; see measured_hit_rate), "
                            f"{a.mode} insert, {a.sessions} sessions/GPU, host keys (pageable int64, the reference's LookupSession::lookup contract), "
                            f"zipf {a.zipf} within the resident set, output rows in HBM",
                "parallelism": "single" if n_rep == 1 else
                               (f"replicas: ONE process, ONE parameter server (host tier built once, {t_tables:.0f} s), one embedding cache + "
                                f"{a.sessions} lookup sessions per GPU on devices {devs}, no data-path collective "
                                "(the reference's arrangement: docs/architecture.md:11,29; model_state.cpp:395-420)"
                                + ("; replicas SHARE devices on this box (fewer GPUs than replicas): each extra replica of a device "
                                   "deploys the model again under its own name" if shared_gpu else "")),
                "host_cpus_per_gpu": ncpu / n_rep,
                "ps_tier": "device-driven (ps_direct_access)" if a.direct else
                           ("host gather" + (f" [{direct_note}]" if direct_note else "")),
                "blocks": blocks, "value_is": "median block", "resident_draw_probability": a.hit,
                "measured_hit_rate": m["measured_hit_rate"],
                "key_bytes_over_pcie": float(np.mean([r[11] for r in main_rec])),
                "timed_region": f"{blocks} blocks of exactly {K} steps per GPU (every device synchronised on both sides; a block ends when "
                                f"the last GPU has finished its {K} steps = max over GPUs), every step a fresh batch; value = the MEDIAN block",
            },
            "block_ms": [b * 1e3 for b in block_s],
            # replica 0's kernels call by call (completion order; us, the kernels' own timestamps) and block by block: does a kernel
            # drift inside the timed region, do a few calls carry the mean (profiles/round6/box_gather_probe_14_boxes.txt)
            "gather_us_per_call": [int(round(r[2] * 1e3)) for r in reps[0].rec],
            "kernel_us_by_block": [[round(float(np.mean([r[c] for r in reps[0].rec[b0:b0 + K]])) * 1e3, 1) for c in (1, 2, 3, 4)]
                                   for b0 in range(0, len(reps[0].rec), K)],
            "value_min_max_over_blocks": [n_rep * K * N / max(block_s), n_rep * K * N / min(block_s)],
            # the spread the median hides: mean / slowest / fastest block, and how many blocks took more than 1.15 x the median
            "value_mean": n_rep * K * N * len(block_s) / sum(block_s),
            "value_min": n_rep * K * N / max(block_s),
            "value_max": n_rep * K * N / min(block_s),
            "slow_blocks": int(sum(1 for b in block_s if b > 1.15 * float(np.median(block_s)))),
            "multi_gpu_selftest": selftest,
            "p50_batch_latency_ms": float(np.percentile(lat, 50)),
            "p99_batch_latency_ms": float(np.percentile(lat, 99)),
            # GPU side of a batch (HIP events on the session's stream: probe start to the last kernel of the call)
            "p50_batch_gpu_ms": float(np.percentile(gpu_ms, 50)),
            "p99_batch_gpu_ms": float(np.percentile(gpu_ms, 99)),
            "measured_hit_rate": m["measured_hit_rate"],
            "keys_narrowed_to_32_bits_fraction_of_calls": float(np.mean([r[10] for r in main_rec])),
            "key_bytes_over_pcie_mean": float(np.mean([r[11] for r in main_rec])),
            "key_stage_ms_mean": m["key_stage_ms"],
            # the five slowest calls of the timed region: [end-to-end ms, ms until the miss counts are on the host,
            # ms inside the host gather calls, ms of upload tail + scatter + insert, ms of the whole call inside the engine]
            "slowest_calls_ms": [[round(l, 3)] + [round(x, 3) for x in p] for l, p in
                                 sorted(zip(lat.tolist(), ph.tolist()), key=lambda t: -t[0])[:5]],
            "host": {"cpus": ncpu, "numa": numa_note, "numa_node_of_worker_pools": hps.pool_numa_node(), "process_bound_to_numa_node": bound_node,
                     "cpu_quota_throttled_periods_in_timed_region": (thr1[0] - thr0[0]) if thr0 and thr1 else None,
                     "cpu_quota_throttled_ms_in_timed_region": (thr1[1] - thr0[1]) / 1e3 if thr0 and thr1 else None,
                     "hypervisor_steal_ms_in_timed_region": (ct1[0] - ct0[0]) * 1e3 if ct0[0] is not None and ct1[0] is not None else None,
                     "process_cpus_busy_in_timed_region": (ct1[1] - ct0[1]) / max(wall1 - wall0, 1e-9)},
            "resident_fraction_after_warmup": resident_frac,
            "per_gpu": per_gpu,
            "roofline": {
                "bound": "hbm",
                # every lookup priced at 1,032 algorithmic bytes over ALL HBM-side kernels of the lookup
                "kernel": "hps_probe_tile_kernel + hps_gather_hits_kernel + hps_miss_scatter_kernel + hps_cache_insert_kernel "
                          "(the probe's tail finds the call-wide unique misses; the insert is enqueued behind the call and counted here)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_return_path_kernels": achieved_rows / HBM_PEAK_GBS,   # probe + gather + scatter: what the caller waits for
                "traffic": traffic,
                # NOT measured in this run (bench.py cannot run the profiler on itself): copied from the committed PMC passes
                "traffic_source": (f"{str(traffic_src).split(' ')[0]} via profiles/pmc_latest.json (not measured in this run)" if traffic_src else None),
                "excluded": "hps_pull_bytes_kernel (keys over PCIe: link-bound, ~4 x 39 us per call), hps_pull16 / hps_push_words (control words)",
                "algorithmic_bytes_per_call": alg,
                "kernel_ms_per_call": hbm_ms,
                "kernel_times": "each kernel's own start/stop timestamps (hipExtLaunchKernel events on the session's stream), averaged over "
                                "the timed region's calls — the launch durations rocprofv3 reports for the same command "
                                "(profiles/round3/ab_kernel_timestamps_vs_event_pairs_vs_rocprofv3.txt: hipEventRecord pairs around the "
                                "launches read 5..8 us more per kernel, the queue hand-offs)",
                "probe_ms": probe, "gather_ms": gather, "scatter_ms": scatter,
                # the cache-insert kernel: since round 4 it is enqueued BEHIND the call (the caller does not wait for it;
                # session option defer_insert 0 puts it back on the return path).  It still occupies the GPU, so the fraction is also given
                # with its time added to the three kernels that produce the call's rows.
                "insert_ms": m["insert_ms"],
                "insert_on_call_path": False,   # (session option defer_insert = 0 puts it back)
                "frac_with_insert": alg / ((hbm_ms + m["insert_ms"]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_probe_plus_gather": alg / ((probe + gather) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # the gather kernel alone on its own bytes (4 B slot per key + 8D per hit) — the dominant kernel
                "frac_gather_own_bytes": (N * 4 + hits * 8 * D) / (gather * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_of_copy_ceiling_6290": achieved_rows / 6290.0,
                # a 1-GiB device-to-device copy on THIS box right before the timed region (read + write bytes / time)
                "box_d2d_copy_GBps": box_copy_gbs,
                "achieved_over_box_copy": achieved_rows / box_copy_gbs if box_copy_gbs else None,
                # the whole job against the same roofline: at 95 % hit with synchronous insertion the path is bound by PCIe
                # (every unique missed row + the keys cross the link once), not by HBM
                "frac_end_to_end": alg / (elapsed / K) / 1e9 / HBM_PEAK_GBS,
                # the kernels with nothing underneath (all keys resident, one session)
                "frac_kernels_alone": (extra.get("all_hit_one_session_device_keys") or {}).get("frac_of_hbm_peak_1032B_per_lookup"),
                # an all-hit call of one session, inside the engine: call time over (probe + gather) kernel time
                "all_hit_call_over_kernel_time": (lambda e: e["engine_call_ms"] / (e["probe_ms"] + e["gather_ms"]) if e else None)(
                    extra.get("all_hit_one_session_device_keys")),
            },
            "roofline_pcie": {
                "bound": "pcie", "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                "bytes_per_step": uniq * 4 * D + N * float(np.mean([r[11] for r in main_rec])),
                "achieved": (uniq * 4 * D + N * float(np.mean([r[11] for r in main_rec]))) / (elapsed / K) / 1e9,
                "frac": (uniq * 4 * D + N * float(np.mean([r[11] for r in main_rec]))) / (elapsed / K) / 1e9 / PCIE_PEAK_GBS,
                "unique_missed_rows_per_batch": uniq,
                "fetch_kernel_ms": fetch_ms if a.direct else None,
                "note": "host->device bytes of one step (unique missed rows + keys) over the step time: the floor of the synchronous path",
            },
            # [1] = wall time of the host gather (host tier) or GPU time of the fetch kernel (device-driven tier)
            "mean_phase_ms": dict(zip(["probe_until_counts_on_host", "ps_fetch", "h2d_scatter_insert", "call"],
                                      [float(x) for x in ph.mean(axis=0)])),
            "extra_legs": extra or None,
            "cpu_baseline": cpu,
            "parity_vs_oracle_bit_exact": parity,
            "parity_full_batch_vs_direct_row_index": parity_full if per_gpu is None or parity_full is None else
                                                     bool(parity_full and all(p.get("parity_full_batch_vs_direct_row_index", True) for p in per_gpu)),
            "checker_note": checker_note,
            "setup_seconds": {"host_tables": t_tables, "gpu_cache_warmup": t_cache},
            "cache_counters": cache.counters(),
        }
    for s in sessions:
        s.close()

    def guarded(name, fn):
        """A leg that can hang (kernels storing over peer mappings, RCCL groups — neither has ever run on this project's
        boxes with more than one GPU) runs on a thread of its own; if it does not come back the line is printed without it."""
        done, box = threading.Event(), {}

        def body():
            try:
                box["r"] = fn()
            except Exception as e:  # noqa: BLE001
                box["r"] = {"error": repr(e)[:300]}
                sys.stderr.write(f"[bench] {name} leg stopped: {e!r}\n")
            done.set()

        threading.Thread(target=body, daemon=True).start()
        if done.wait(a.sharded_timeout):
            return box["r"]
        try:   # where is it stuck?  (every Python thread's stack, to stderr)
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        except Exception:  # noqa: BLE001
            pass
        res["extra_legs"] = dict(res.get("extra_legs") or {}, **{name: {"error": f"no result within {a.sharded_timeout} s"}})
        emit(res)
        os._exit(0)


    # ---- one GPU: legs that need the headline's memory back (its tables are 133 GB): the plugin boundary driven by the
    #      native load generator, then the other parameter-server tier ----
    if n_rep == 1 and not a.no_extra_legs:
        import gc
        del sessions, cache, caches, ps, made, run, host_batches, reps
        gc.collect()
        torch.cuda.empty_cache()
        if not a.no_triton_leg:
            res["extra_legs"] = dict(res["extra_legs"] or {}, triton_abi=triton_abi_leg(a, hb, T, R, D, B),
                                     c1_cpu_ps_triton=c1_leg(a, hb), c4_two_models_triton=c4_leg(a, hb))
        if not a.no_wide_leg:
            try:
                wleg = wide_keys_leg(a, torch, hps, T, R, D, B, N, dev, cfg)
            except Exception as e:  # noqa: BLE001
                wleg = {"error": repr(e)[:300]}
                sys.stderr.write(f"[bench] wide-keys leg stopped: {e!r}\n")
            res["extra_legs"] = dict(res["extra_legs"] or {}, wide_keys_95=wleg)
            gc.collect()
            torch.cuda.empty_cache()
        if not a.direct and not a.no_direct_leg:
            try:
                dleg = other_tier_leg(a, torch, hps, T, R, D, B, N, dev, cfg)
            except Exception as e:  # noqa: BLE001
                dleg = {"error": repr(e)[:300]}
                sys.stderr.write(f"[bench] device-driven tier leg stopped: {e!r}\n")
            res["extra_legs"] = dict(res["extra_legs"] or {}, device_driven_tier=dleg)
        if not a.no_c3_leg:
            gc.collect()
            torch.cuda.empty_cache()
            c3 = guarded("sharded_c3_logical", lambda: c3_logical_leg(a, torch, hps, dev))
            res["extra_legs"] = dict(res["extra_legs"] or {}, sharded_c3_logical=c3)
            gc.collect()
            torch.cuda.empty_cache()
            c3e = guarded("sharded_c3_single_entry", lambda: c3_single_entry_leg(a, torch, hps, [dev] * 4, 1 << 24))
            res["extra_legs"] = dict(res["extra_legs"] or {}, sharded_c3_single_entry=c3e)
            if not a.no_triton_leg:
                res["extra_legs"] = dict(res["extra_legs"] or {}, c3_sharded_triton=c3_triton_leg(a, hb, 4))

    # ---- BASELINE config 3 leg (only under torch.distributed.run with N > 1 ranks): ONE table sharded over the ranks, one
    # rank per GPU, RCCL send/recv inside the engine.  Runs after the headline measurement is complete and its resources are
    # released (the barrier below is what the idle ranks have been waiting at); a watchdog prints the headline line and ends
    # the rank if the leg does not finish (a collective that hangs cannot be caught any other way).
    if n_rep > 1:
        import gc
        del sessions, cache, caches, ps, made, run, host_batches, reps
        gc.collect()
        for d in sorted(set(devs)):
            with torch.cuda.device(d):
                torch.cuda.empty_cache()
    if n_rep > 1 and not a.no_extra_legs and not a.no_c3_leg:
        # ---- BASELINE config 3 on the N GPUs of this process, both variants: (a) behind the plugin's contract — a table-sharded
        # model whose entry instances serve whole requests, the owners writing rows over peer mappings (shard_entry.h); (b) the
        # SPMD session over RCCL with one rank per device, the ranks being threads of this process.  Fewer devices than
        # replicas (development box): (a) runs with logical shards on the devices there are, (b) with as many RCCL ranks as
        # there are devices.
        entry_rows = min(a.shard_rows, 1 << 26) if not shared_gpu else 1 << 24

        c3e = guarded("sharded_c3_single_entry", lambda: c3_single_entry_leg(a, torch, hps, devs, entry_rows, transports=c3_transports))
        res["extra_legs"] = dict(res.get("extra_legs") or {}, sharded_c3_single_entry=c3e)
        if not a.no_triton_leg:
            res["extra_legs"] = dict(res.get("extra_legs") or {}, c3_sharded_triton=c3_triton_leg(a, hb, n_rep))
        gc.collect()
        if world == 1 and not rccl_usable:
            res["extra_legs"] = dict(res.get("extra_legs") or {}, sharded_c3={"error": "skipped: the self-test's RCCL all-reduce did not come back"})
        elif world == 1:
            ranks = min(n_rep, ndev)
            c3r = guarded("sharded_c3", lambda: c3_rccl_threads_leg(a, torch, hps, ranks, entry_rows))
            res["extra_legs"] = dict(res.get("extra_legs") or {}, sharded_c3=c3r)
    if world > 1:
        dist.barrier()
        if not a.no_sharded_leg:
            def give_up():
                res["extra_legs"] = dict(res.get("extra_legs") or {}, sharded_c3={"error": f"no result within {a.sharded_timeout} s"})
                emit(res)
                os._exit(0)

            dog = threading.Timer(a.sharded_timeout, give_up)
            dog.daemon = True
            dog.start()
            try:
                leg3 = sharded_leg(a, torch, dist, hps, rank, world, local_rank % ndev, world > ndev)
            except Exception as e:  # noqa: BLE001
                leg3 = {"error": repr(e)[:300]}
            dog.cancel()
            res["extra_legs"] = dict(res.get("extra_legs") or {}, sharded_c3=leg3)
    emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    silence_stdout_for_exit()


def silence_stdout_for_exit():
    """Nothing may follow the result line on stdout, and libraries print when they are torn down (RCCL's version banner; under
    torch.distributed.run every rank's stdout is merged with rank 0's).  The process still has to exit the normal way — a
    profiler's tool library (rocprofv3) writes its results from its exit handlers; round 5's first answer, os._exit, left
    `rocprofv3 -- python bench.py` without an output directory — so file descriptor 1 is pointed at /dev/null instead."""
    sys.stdout.flush()
    sys.stderr.flush()
    try:
        ctypes.CDLL(None).fflush(None)
        dn = os.open(os.devnull, os.O_WRONLY)
        os.dup2(dn, 1)
        os.close(dn)
    except Exception:  # noqa: BLE001
        pass


def idle_rank(a, dist, rank, world, local_rank):
    """Ranks 1..N-1 under torch.distributed.run.  The replicas measurement is rank 0's alone (one process serves every GPU,
    as the reference does), so these ranks wait for it at a barrier; afterwards each takes its own GPU for the sharded-table
    leg (BASELINE config 3), where one process per GPU is the arrangement."""
    dist.barrier()
    if not a.no_sharded_leg:
        dog = threading.Timer(a.sharded_timeout, lambda: os._exit(0))
        dog.daemon = True
        dog.start()
        try:
            from hugectr_backend_amd.gpu_wait import wait_for_gpu
            wait_for_gpu(30.0)
            import torch
            from hugectr_backend_amd import hps
            ndev = torch.cuda.device_count()
            sharded_leg(a, torch, dist, hps, rank, world, local_rank % ndev, world > ndev)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[bench rank {rank}] sharded leg stopped: {e!r}\n")
        dog.cancel()
    dist.barrier()
    dist.destroy_process_group()
    silence_stdout_for_exit()


def run_abi_driver(hb, args, timeout):
    """tools/triton_abi_bench.cpp (native): plays tritonserver through the mock core and drives
    TRITONBACKEND_ModelInstanceExecute of libtriton_hps.so — what perf_analyzer does to the reference (.gitlab-ci.yml:70).
    Own process: it loads its models through TRITONBACKEND_ModelInitialize itself."""
    import subprocess
    exe = hb.LIB / "triton_abi_bench.bin"
    if not exe.exists():
        return {"error": "tools/triton_abi_bench.cpp was not built"}
    try:
        r = subprocess.run([str(exe), "--lib-dir", str(hb.LIB), *map(str, args)], capture_output=True, text=True, timeout=timeout)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            return {"error": f"rc={r.returncode}: {r.stderr[-300:]}"}
        out = json.loads(line[-1])
        out["rc"] = r.returncode
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def triton_abi_leg(a, hb, T, R, D, B):
    """The headline workload through the plugin boundary (two instances, pageable KEYS, device OUTPUT0)."""
    return run_abi_driver(hb, ["--tables", T, "--rows", R, "--dim", D, "--batch", B, "--cache-frac", a.cache_frac, "--hit", a.hit,
                               "--zipf", a.zipf, "--instances", a.sessions, "--steps", 20, "--blocks", 12, "--warmup", 5,
                               "--direct", int(bool(a.direct)), "--also-pinned", 4], a.triton_timeout)


def c3_triton_leg(a, hb, shards):
    """BASELINE config 3 THROUGH THE PLUGIN: one table-sharded model (ps.json "table_sharding": "hash"), `shards` shards on the visible
    devices (shard s on device s mod ndev), one instance per device of the pool, requests of the headline's size (1,703,936 uniform
    keys of one table of 2^24 rows, 100 % resident) through TRITONBACKEND_ModelInstanceExecute — every instance is an entry session
    (csrc/cache/shard_entry.h).  Native driver, pageable KEYS, device OUTPUT0."""
    N = a.tables * a.batch
    out = run_abi_driver(hb, ["--tables", 1, "--rows", 1 << 24, "--dim", a.dim, "--batch", N, "--cache-frac", 1.0, "--uniform", 1,
                              "--shards", shards, "--instances", 2, "--steps", 20, "--blocks", 4, "--warmup", 4], 180.0)
    out["config"] = (f"BASELINE configs[2] through the plugin: one table of 2^24 rows x {a.dim} sharded over {shards} shards "
                     f"(table_sharding hash), requests of {N} uniform keys, two instances per device")
    return out


def c1_leg(a, hb):
    """BASELINE config 1 — the reference's own CI performance smoke (.gitlab-ci.yml:70: perf_analyzer against the hps backend):
    one table 1,048,576 x 16, 4,096-key requests, CPU parameter server only (gpucache=false, KIND_CPU instance, OUTPUT0 in host
    memory), through TRITONBACKEND_ModelInstanceExecute."""
    out = run_abi_driver(hb, ["--tables", 1, "--rows", 1 << 20, "--dims", 16, "--batch", 4096, "--gpucache", 0, "--uniform", 1,
                              "--instances", 1, "--steps", 500, "--blocks", 6, "--warmup", 200], 120.0)
    out["config"] = "BASELINE configs[0]: 1 table 1,048,576 x 16 fp32, 4,096 uniform keys per request, gpucache=false, one KIND_CPU instance"
    return out


def c4_leg(a, hb):
    """BASELINE config 4 — two W&D models (README.md:148-152: D = [1,16], keys per sample [2,26]) served side by side on one GPU,
    batch 1,024 = 28,672 keys per request, one instance each, synchronous insertion (exact rows), hit rate swept."""
    res = {}
    for hit in (0.5, 0.9, 0.99):
        res[f"target_hit_{hit}"] = run_abi_driver(
            hb, ["--models", 2, "--dims", "1,16", "--per-sample", "2,26", "--rows", 1_000_000, "--batch", 1024, "--instances", 1,
                 "--cache-frac", 0.2, "--hit", hit, "--zipf", a.zipf, "--steps", 100, "--blocks", 6, "--warmup", 50,
                 "--direct", int(bool(a.direct))], 120.0)
    # the same requests as a dynamic batcher hands them over: EIGHT per TRITONBACKEND_ModelInstanceExecute call (max_batch_size 8,192).
    # The reference runs one blocking lookup per request (hps.cc:406); here the eight go as one engine call (csrc/triton/hps.cpp).
    batched = run_abi_driver(
        hb, ["--models", 2, "--dims", "1,16", "--per-sample", "2,26", "--rows", 1_000_000, "--batch", 1024, "--instances", 1,
             "--cache-frac", 0.2, "--hit", 0.9, "--zipf", a.zipf, "--steps", 40, "--blocks", 6, "--warmup", 20,
             "--direct", int(bool(a.direct)), "--requests-per-execute", 8], 120.0)
    return {"config": "BASELINE configs[3]: two W&D models (2 tables each, 1,000,000 rows x [1,16] fp32, keys/sample [2,26], batch 1,024 = "
                      "28,672 keys per request), one GPU instance each, concurrent Execute, sync insert, pageable KEYS, device OUTPUT0",
            "results": res, "batched8_target_hit_0.9": batched}


def fresh_deployment_leg(a, torch, hps, T, R, D, B, N, dev, cfg, direct, key0=0, check_rows=False, more=None, narrow_keys=None):
    """The headline workload on a deployment of its own (own server, the headline's has been released): two sessions, fresh
    batches of host keys every step, exact rows.  direct: the parameter-server tier; key0: the tables' keys are
    key0 .. key0+R-1 (key0 = 2^40: keys that need all 8 bytes over PCIe — the reference takes int64 keys, hps.cc:573)."""
    model = cfg["models"][0]["model"]
    cfg = json.loads(json.dumps(cfg))
    cfg["models"] = cfg["models"][:1]
    cfg["models"][0]["ps_direct_access"] = bool(direct)
    cfg["models"][0]["deployed_device_list"] = [dev]
    t0 = time.time()
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t in range(T):
        ps.load_table_synthetic(model, t, SEED, key0, R)
    ps.create_embedding_cache_per_model(model)
    cache = ps.get_embedding_cache(model, dev)
    t_setup = time.time() - t0
    sessions = [hps.LookupSession.create(ps, model, cache) for _ in range(a.sessions)]
    for s in sessions:
        s.set_option("timing", 1)
        s.set_option("probe_variant", a.probe_variant)
        s.set_option("xcd_walk", a.xcd_walk)
        s.set_option("narrow_keys", a.narrow_keys if narrow_keys is None else narrow_keys)
        s.set_option("split_probe", 1 if (a.split_probe != 0 and not direct) else 0)
    C = int(np.ceil(a.cache_frac * R))
    resident = []
    for t in range(T):
        k = np.arange(C, dtype=np.int64)
        resident.append(torch.from_numpy(k[cache.query(t, k + key0) >= 0]).cuda())
    gen = torch.Generator(device="cuda")
    gen.manual_seed(SEED + 555 + (key0 & 0xFFFF) + (key0 >> 32))
    cdf_d = torch.from_numpy(zipf_cdf(C, a.zipf)).cuda()
    run = Runner(torch, hps, sessions, T, B, D, dev)
    steps = 40
    hb_ = [((x + key0).cpu().numpy(),) for x in make_batches_gpu(torch, gen, resident, cdf_d, R, C, B, a.hit, steps + 8)]
    hb_ = [(x[0], run.pack_host(x[0])) for x in hb_]
    rec = []
    run.run(hb_, 8, 0, "host")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run.run(hb_, steps, 8, "host", record=rec)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    out = summarize(rec, N, D, dt, steps)
    out.update({"sessions": len(sessions), "steps": steps, "setup_seconds": t_setup,
                "key_bytes_over_pcie_mean": float(np.mean([r[11] for r in rec]))})
    if check_rows:
        # the last batch again, every row against the table as it sits in the host tier (file order = key order: row of key k
        # is row k - key0)
        q = hb_[-1][0]
        sessions[0].lookup_packed(hb_[-1][1], run.vptrs[0], run.counts)
        torch.cuda.synchronize()
        got = run.outs[0].cpu().numpy().reshape(N, D)
        ok = True
        for t in range(T):
            tk, tr = ps.table_data(model, t)
            qt = q[t * B:(t + 1) * B]
            ok &= bool(np.array_equal(tk[qt - key0], qt))
            ok &= bool(np.array_equal(tr[qt - key0].view(np.uint32), got[t * B:(t + 1) * B].view(np.uint32)))
        out["parity_full_batch_vs_direct_row_index"] = ok
    if more is not None:
        try:   # further legs on this deployment while it is up
            more(out, dict(ps=ps, cache=cache, sessions=sessions, run=run, resident=resident, cdf_d=cdf_d, gen=gen, C=C, model=model,
                           key0=key0))
        except Exception as e:  # noqa: BLE001
            out["more_error"] = repr(e)[:300]
            sys.stderr.write(f"[bench] leg on the fresh deployment stopped: {e!r}\n")
        finally:
            run.post_hooks[:] = []
            run.step_hooks[:] = []
    for s in sessions:
        s.close()
    return out, rec


def other_tier_leg(a, torch, hps, T, R, D, B, N, dev, cfg):
    """The headline workload on a ps_direct_access deployment of the same model (own server: page-locked tables, device
    index): two sessions, fresh batches of host keys, exact rows."""
    def c5_fused(out, ctx):
        """BASELINE config 5 on this deployment: lookup + bottom MLP + interaction as separate steps (OUTPUT0 written and read
        back) against the fused call (hps_session_lookup_interact_device: the interaction reads cache slots / staged rows,
        OUTPUT0 never exists); device keys, exact rows, same batches for both, outputs compared bit for bit."""
        if not (D % 32 == 0 and D <= 512 and T <= 31 and a.mode == "sync"):
            return
        from hugectr_backend_amd.dense import DenseInteraction
        sessions, run = ctx["sessions"], ctx["run"]
        rngw = np.random.default_rng(SEED)
        dims, k = [512, 256, D], 13
        ws, bs = [], []
        for n_ in dims:
            ws.append(((rngw.random((k, n_), dtype=np.float32) * 2 - 1) * (1.5 / np.sqrt(k))).astype(np.float32))
            bs.append(((rngw.random(n_, dtype=np.float32) - 0.3) * 0.2).astype(np.float32))
            k = n_
        ops = [DenseInteraction(ws, bs, T, D, device=dev) for _ in sessions]
        xd = torch.randn(B, 13, device="cuda")
        outd = [torch.empty((B, ops[0].out_stride), dtype=torch.float16, device="cuda") for _ in sessions]

        def fresh(n_, hit=None):
            return make_batches_gpu(torch, ctx["gen"], ctx["resident"], ctx["cdf_d"], R, ctx["C"], B, a.hit if hit is None else hit, n_)

        def leg(batches, steps):
            r = []
            run.run(batches, 4, 0, "device")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.run(batches, steps, 4, "device", record=r)
            torch.cuda.synchronize()
            return summarize(r, N, D, time.perf_counter() - t0, steps)

        res = {}
        for name, hit in (("hit_95", None), ("all_hit", 1.1)):
            bt = fresh(28, hit)
            run.step_hooks[:] = []
            run.post_hooks[:] = [lambda si: (ops[si].forward(xd, run.outs[si], B, out=outd[si]),
                                             torch.cuda.current_stream().synchronize()) for _ in sessions]
            un = leg(bt, 24)
            ref = outd[0].clone()
            last = bt[(4 + 23) % len(bt)]
            # which batch session 0 saw last is not fixed (two sessions share the steps): recompute one batch on session 0 for the check
            sessions[0].lookup_device(last, run.nk, out=run.outs[0])
            ops[0].forward(xd, run.outs[0], B, out=outd[0])
            torch.cuda.synchronize()
            ref = outd[0].clone()
            run.post_hooks[:] = []
            run.step_hooks[:] = [lambda si, keys: ops[si].lookup_interact(sessions[si], keys, B, xd, out=outd[si]) for _ in sessions]
            # session option "interact_mode": 1 = always the fused arrangement (round 5's only one), 2 = the default: fused while
            # the session's calls miss little, lookup + dense steps inside the call while they miss much
            for s_ in sessions:
                s_.set_option("interact_mode", 1)
            fa = leg(fresh(28, hit), 24)
            for s_ in sessions:
                s_.set_option("interact_mode", 2)
            fu = leg(fresh(28, hit), 24)
            sep_calls = int(sum(s_.last_stats().interact_separate for s_ in sessions))
            ops[0].lookup_interact(sessions[0], last, B, xd, out=outd[0])
            torch.cuda.synchronize()
            same = bool(torch.equal(ref, outd[0]))
            run.step_hooks[:] = []
            res[name] = {"separate_steps_samples_per_s": un["lookups_per_s"] / T, "separate_ms_per_step": un["ms_per_step"],
                         "fused_always_ms_per_step": fa["ms_per_step"], "sessions_whose_last_call_ran_the_separate_steps": sep_calls,
                         "fused_samples_per_s": fu["lookups_per_s"] / T, "fused_ms_per_step": fu["ms_per_step"],
                         "fused_over_separate": fu["lookups_per_s"] / un["lookups_per_s"],
                         "fused_p50_call_ms": fu["p50_call_ms"], "fused_p99_call_ms": fu["p99_call_ms"], "fused_max_call_ms": fu["max_call_ms"],
                         "outputs_bit_identical": same}
        out["c5_fused_lookup_interact"] = dict(res, note="BASELINE configs[4]: lookup + bottom MLP 13-512-256-%d + dot interaction per "
                                               "step, device keys, two sessions; separate = OUTPUT0 written then read by the dense "
                                               "kernels, fused = one call whose interaction kernel reads cache slots / staged rows" % D)

    out, rec = fresh_deployment_leg(a, torch, hps, T, R, D, B, N, dev, cfg, direct=True, more=c5_fused)
    f_ms = float(np.mean([r[8][1] for r in rec]))
    uniq = float(np.mean([r[6] for r in rec]))
    out.update({
        "roofline_pcie": {"bound": "pcie", "kernel": "hps_ps_fetch_direct_kernel", "avg_kernel_ms": f_ms,
                          "achieved": uniq * 4 * D / (f_ms * 1e-3) / 1e9 if f_ms > 0 else None, "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                          "frac": uniq * 4 * D / (f_ms * 1e-3) / 1e9 / PCIE_PEAK_GBS if f_ms > 0 else None},
        "note": "same workload (host keys), two sessions, exact rows; the GPU resolves the misses through a device index of the "
                "page-locked host tables and reads the rows over PCIe itself (no host threads on the miss path)",
    })
    return out


def wide_keys_leg(a, torch, hps, T, R, D, B, N, dev, cfg):
    """The headline workload with keys that need all 64 bits on the wire: tables keyed 2^40 .. 2^40+R-1, same recipe, same
    tier and session options as the headline, 95 % hit, every step a fresh batch.  Two measurements on the one deployment:
    the keys sent as they are (narrowing off: what ids without structure — hashed 64-bit ids — cost), then with the engine's
    default, which narrows them as offsets from each table's smallest key (frame of reference, csrc/cache/key_pack.h)."""
    key0 = 1 << 40

    def with_default_narrowing(out, ctx):
        run, sessions = ctx["run"], ctx["sessions"]
        for s in sessions:
            s.set_option("narrow_keys", a.narrow_keys)
            s.set_option("keys_by_kernel", a.keys_by_kernel)
        steps = 40
        hb_ = [((x + key0).cpu().numpy(),) for x in make_batches_gpu(torch, ctx["gen"], ctx["resident"], ctx["cdf_d"], R, ctx["C"], B, a.hit, steps + 8)]
        hb_ = [(x[0], run.pack_host(x[0])) for x in hb_]
        rec = []
        run.run(hb_, 8, 0, "host")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run.run(hb_, steps, 8, "host", record=rec)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        o = summarize(rec, N, D, dt, steps)
        o["key_bytes_over_pcie_mean"] = float(np.mean([r[11] for r in rec]))
        q = hb_[-1][0]
        sessions[0].lookup_packed(hb_[-1][1], run.vptrs[0], run.counts)
        torch.cuda.synchronize()
        got = run.outs[0].cpu().numpy().reshape(N, D)
        ok = True
        for t in range(T):
            tk, tr = ctx["ps"].table_data(ctx["model"], t)
            qt = q[t * B:(t + 1) * B]
            ok &= bool(np.array_equal(tr[qt - key0].view(np.uint32), got[t * B:(t + 1) * B].view(np.uint32)))
        o["parity_full_batch_vs_direct_row_index"] = ok
        o["note"] = ("the engine's default on the same deployment: the keys narrowed as offsets from each table's smallest key "
                     "(3 bytes each here: the tables' ids are dense above 2^40)")
        out["frame_of_reference_default"] = o

    def with_pinned_keys(out, ctx):
        # the same wide keys in PAGE-LOCKED host arrays (Triton's pinned input pool; libtriton_hps.so asks for it): DMA in place at
        # 8 bytes per key, no staging copy on the host
        run, sessions = ctx["run"], ctx["sessions"]
        steps = 40
        raw = [(x + key0).cpu() for x in make_batches_gpu(torch, ctx["gen"], ctx["resident"], ctx["cdf_d"], R, ctx["C"], B, a.hit, steps + 8)]
        pinned = [x.pin_memory() for x in raw]
        del raw
        hb_ = [(p_, run.pack_host(p_.numpy())) for p_ in pinned]
        rec = []
        run.run(hb_, 8, 0, "pinned")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run.run(hb_, steps, 8, "pinned", record=rec)
        torch.cuda.synchronize()
        o = summarize(rec, N, D, time.perf_counter() - t1, steps)
        o["key_bytes_over_pcie_mean"] = float(np.mean([r[11] for r in rec]))
        o["note"] = "the wide keys in page-locked host arrays: DMA'd in place, 8 bytes per key, never read by a host thread"
        out["pinned_keys_dma_in_place"] = o

    def both(out, ctx):
        with_pinned_keys(out, ctx)
        with_default_narrowing(out, ctx)

    out, rec = fresh_deployment_leg(a, torch, hps, T, R, D, B, N, dev, cfg, direct=bool(a.direct), key0=key0, check_rows=True,
                                    more=both, narrow_keys=0)
    uniq = float(np.mean([r[6] for r in rec]))
    bytes_step = uniq * 4 * D + N * out["key_bytes_over_pcie_mean"]
    out.update({
        "key0": key0,
        "pcie_bytes_per_step": bytes_step,
        "pcie_GBps": bytes_step / (out["ms_per_step"] * 1e-3) / 1e9,
        "pcie_frac_of_63": bytes_step / (out["ms_per_step"] * 1e-3) / 1e9 / PCIE_PEAK_GBS,
        "note": "keys 2^40 + (the headline's key distribution) sent as they are (session option narrow_keys 0): KEYS cross PCIe at "
                "8 bytes each (13.6 MB per step next to ~37 MB of missed rows) — what ids without structure cost; the step is "
                "PCIe-bound, so the floor is the headline's time x (rows + 8-byte keys) / (rows + 3-byte keys).  "
                "frame_of_reference_default: the same traffic with the engine's default narrowing",
    })
    return out


def c3_logical_leg(a, torch, hps, dev, P=4, rows_total=1 << 24, steps=30):
    """BASELINE configs[2] on the ONE GPU of this box: one table sharded over P LOGICAL shards (owner = mix64(key) mod P), P
    servers + caches + lookup sessions + native sharded sessions in this process, P host threads — the production path of
    csrc/cache/shard_session.cpp (input dedup, bucket into fixed-capacity blocks, exchange, padded local lookup, exchange,
    gather back) with device-to-device copies standing in for RCCL's send/recv groups.  What it times: the bucket / prepare /
    padded-lookup / gather-back kernels and the session's control flow.  What it does NOT time: RCCL and xGMI — no multi-GPU node
    was in reach of this project (SCALE_r01..r03: skipped)."""
    import ctypes as C
    from oracle import hps_oracle as O
    D = a.dim
    N = a.tables * a.batch
    n_local = N // P
    recv_cap = int(n_local * 1.25) + 4096
    t0 = time.time()
    servers, sessions, shards = [], [], []
    grp = C.c_void_p()
    hps._check(hps.LIB.hps_shard_group_create_local(P, C.byref(grp)))
    for r in range(P):
        model = f"c3_logical_{r}"
        cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
               "models": [{"model": model, "sparse_files": ["synthetic://shard"], "num_of_worker_buffer_in_pool": 2,
                           "embedding_vecsize_per_table": [D], "maxnum_catfeature_query_per_table_per_sample": [1],
                           "default_value_for_each_table": [0.0], "deployed_device_list": [dev], "max_batch_size": recv_cap,
                           "gpucache": True, "gpucacheper": 1.0, "hit_rate_threshold": 1.0}]}
        ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
        ps.load_table_synthetic(model, 0, SEED, 0, rows_total, shard=r, num_shards=P)
        ps.create_embedding_cache_per_model(model)
        sess = hps.LookupSession.create(ps, model, ps.get_embedding_cache(model, dev))
        h = C.c_void_p()
        hps._check(hps.LIB.hps_shard_session_create_local(sess._h, grp, r, n_local, C.byref(h)))
        servers.append(ps), sessions.append(sess), shards.append(h)
    t_setup = time.time() - t0
    rng = np.random.default_rng(SEED + 303)
    zw = 1.0 / np.power(np.arange(1, 1_000_001, dtype=np.float64), a.zipf)
    zw /= zw.sum()
    zipf_ids = rng.permutation(rows_total)[:1_000_000].astype(np.int64)

    def batches(kind):
        if kind == "uniform":
            return [[rng.integers(0, rows_total, n_local, dtype=np.int64) for _ in range(4)] for _ in range(P)]
        return [[zipf_ids[rng.choice(zipf_ids.size, n_local, p=zw)] for _ in range(4)] for _ in range(P)]

    outs = [torch.empty(n_local * D, dtype=torch.float32, device=torch.device("cuda", dev)) for _ in range(P)]
    res = {"shards": P, "rows_total": rows_total, "keys_per_step": N, "keys_per_rank_and_step": n_local, "setup_seconds": t_setup,
           "transport": "in-process (device-to-device copies on one GPU) — RCCL / xGMI NOT measured: no multi-GPU node in reach",
           "note": c3_logical_leg.__doc__.split("\n")[0]}
    for kind, warm in (("uniform", 4), ("zipf", 40)):     # (zipf: 32 calls let the block capacity follow the deduplicated traffic down)
        bt = batches(kind)
        bar = threading.Barrier(P + 1)
        errs, tim, stat = [], [[] for _ in range(P)], [None] * P

        def work(r):
            try:
                torch.cuda.set_device(dev)
                for i in range(warm):
                    hps._check(hps.LIB.hps_shard_session_lookup_host(shards[r], bt[r][i % 4].ctypes.data, n_local, outs[r].data_ptr()))
                bar.wait()
                bar.wait()
                for i in range(steps):
                    hps._check(hps.LIB.hps_shard_session_lookup_host(shards[r], bt[r][i % 4].ctypes.data, n_local, outs[r].data_ptr()))
                    t = [C.c_float(0) for _ in range(3)]
                    recv, kb = C.c_uint64(0), C.c_int32(0)
                    hps._check(hps.LIB.hps_shard_session_last_timing(shards[r], C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), C.byref(recv), C.byref(kb)))
                    tim[r].append([x.value for x in t] + [recv.value, kb.value])
                att, cap = C.c_uint32(0), C.c_uint64(0)
                sent = (C.c_uint64 * P)()
                hps._check(hps.LIB.hps_shard_session_last_stats(shards[r], C.byref(cap), C.byref(att), sent, P))
                stat[r] = (att.value, cap.value, int(sum(sent)))
                bar.wait()
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e)[:200])
                bar.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
        [x.start() for x in th]
        try:
            bar.wait()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            bar.wait()
            bar.wait()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
        except threading.BrokenBarrierError:
            dt = float("nan")
        [x.join() for x in th]
        if errs:
            res[kind] = {"error": errs[0]}
            continue
        # parity: every rank's last answer against the row recipe, 256 sampled positions each
        ok = True
        for r in range(P):
            kh = bt[r][(steps - 1) % 4]
            idx = np.linspace(0, n_local - 1, 256).astype(np.int64)
            exp = np.concatenate([O.c_synth_rows(SEED, 0, int(kh[i]), 1, D) for i in idx]).reshape(256, D)
            got = outs[r].view(-1, D)[torch.from_numpy(idx).to(outs[r].device)].cpu().numpy()
            ok &= bool(np.array_equal(got.view(np.uint32), exp.view(np.uint32)))
        tm = np.mean(np.array([x for r in range(P) for x in tim[r]], dtype=np.float64), axis=0)
        cap = stat[0][1]
        res[kind] = {"lookups_per_s": N * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "parity": ok,
                     "keys_exchange_ms": float(tm[0]), "local_lookup_ms": float(tm[1]), "rows_exchange_ms": float(tm[2]),
                     "key_bytes_over_pcie": float(tm[4]), "block_capacity_keys": cap, "attempts_last_step": max(s_[0] for s_ in stat),
                     "distinct_keys_sent_per_rank": float(np.mean([s_[2] for s_ in stat])),
                     "row_bytes_per_peer_and_step": cap * 4 * D,
                     "padding_fraction_of_row_blocks": 1.0 - float(np.mean([s_[2] for s_ in stat])) / P / cap}
    res["lookups_per_s"] = (res.get("uniform") or {}).get("lookups_per_s")
    res["parity"] = bool((res.get("uniform") or {}).get("parity") and (res.get("zipf") or {}).get("parity"))
    for h in shards:
        hps.LIB.hps_shard_session_destroy(h)
    hps.LIB.hps_shard_group_destroy(grp)
    for s_ in sessions:
        s_.close()
    return res


def c3_single_entry_leg(a, torch, hps, shard_devs, rows_total, steps=30, transports=("peer_store", "staged_copy")):
    """BASELINE configs[2] BEHIND THE PLUGIN'S CONTRACT (one blocking call per request on one instance, hps.cc:353-369): one
    table-sharded model (ps.json "table_sharding": "hash", csrc/cache/shard_entry.h) — ONE server, the host tier whole, shard s
    = entry s of deployed_device_list with 100 % of the keys it owns resident; an ENTRY session per instance buckets a request of
    the headline's size by owner on its device, drives P lookup sessions (one per shard, on the shard's device) from its own
    threads, and the shards' gather kernels store the rows straight into the entry's output over peer mappings.  No collective.
    Timed: one instance alone (host keys, then device keys; uniform, then Zipf keys), then a request on EVERY instance at once."""
    import ctypes as C
    from oracle import hps_oracle as O
    P, D = len(shard_devs), a.dim
    N = a.tables * a.batch
    model = "criteo_c3_entry"
    t0 = time.time()
    cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
           "models": [{"model": model, "sparse_files": ["synthetic://one_table"], "num_of_worker_buffer_in_pool": max(2, P),
                       "embedding_vecsize_per_table": [D], "maxnum_catfeature_query_per_table_per_sample": [1],
                       "default_value_for_each_table": [0.0], "deployed_device_list": list(shard_devs), "max_batch_size": N,
                       "gpucache": True, "gpucacheper": 1.0, "gpucache_load_factor": float(os.environ.get("ENTRY_LOAD_FACTOR", "0.5")),
                       "hit_rate_threshold": 1.0, "table_sharding": "hash",
                       "shard_transport": transports[0], "shard_copy_piece_keys": a.copy_piece_keys}]}
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    ps.load_table_synthetic(model, 0, SEED, 0, rows_total)
    t_table = time.time() - t0
    ps.create_embedding_cache_per_model(model)
    t_cache = time.time() - t0 - t_table
    entry_devs = list(shard_devs)          # one instance per deployed entry (Triton: instance_group count per GPU)
    entries = [hps.ShardedEntrySession.create(ps, model, d) for d in entry_devs]
    rng = np.random.default_rng(SEED + 404)
    zw = 1.0 / np.power(np.arange(1, 1_000_001, dtype=np.float64), a.zipf)
    zw /= zw.sum()
    zipf_ids = rng.permutation(rows_total)[:1_000_000].astype(np.int64)

    def batches(kind, count):
        if kind == "uniform":
            return [rng.integers(0, rows_total, N, dtype=np.int64) for _ in range(count)]
        return [zipf_ids[rng.choice(zipf_ids.size, N, p=zw)] for _ in range(count)]

    outs = [torch.empty(N * D, dtype=torch.float32, device=torch.device("cuda", d)) for d in entry_devs]
    distinct_devs = sorted(set(shard_devs))
    res = {"shards": P, "shard_devices": list(shard_devs), "rows_total": rows_total, "keys_per_request": N,
           "setup_seconds": {"host_table": t_table, "shard_caches": t_cache}, "shard_capacity_keys": entries[0].shard_capacity,
           "transport": ("peer-mapped stores over xGMI (owners write into the entry GPU's output); no collective" if len(distinct_devs) > 1 else
                         "ALL SHARDS ON ONE GPU (logical shards): the owners' stores stay in local HBM — xGMI NOT measured"),
           "note": c3_single_entry_leg.__doc__.split("\n")[0]}

    def sync_all():
        for d in distinct_devs:
            torch.cuda.synchronize(d)

    def check(e, out, kh):
        idx = np.linspace(0, N - 1, 256).astype(np.int64)
        exp = np.concatenate([O.c_synth_rows(SEED, 0, int(kh[i]), 1, D) for i in idx]).reshape(256, D)
        got = out.view(-1, D)[torch.from_numpy(idx).to(out.device)].cpu().numpy()
        return bool(np.array_equal(got.view(np.uint32), exp.view(np.uint32)))

    ok_all = True
    # Both transports of the rows on the SAME requests (csrc/cache/shard_entry.h): "peer_store" — the owners' kernels store into the
    # entry GPU's output over peer mappings — with uniform and Zipf keys, then "staged_copy" — owners gather pieces into local
    # blocks, hipMemcpyPeerAsync ships them, a kernel on the entry GPU puts the rows in place — with the uniform requests.
    # `transports`: what the self-test left standing (a machine without peer access, or whose peer stores hung, runs staged_copy only).
    plan = [(tr, kind) for tr in transports for kind in (("uniform", "zipf") if tr == transports[0] else ("uniform",))]
    saved = {}
    for tr, kind in plan:
        for e_ in entries:
            e_.set_option("transport", 1 if tr == "staged_copy" else 0)
        if kind not in saved:
            saved[kind] = batches(kind, 4)
        bt = saved[kind]
        e, out = entries[0], outs[0]
        for i in range(4):
            e.lookup(bt[i % 4], [N], out=out)
        sync_all()
        lat, ph = [], []
        t1 = time.perf_counter()
        for i in range(steps):
            ts = time.perf_counter()
            e.lookup(bt[i % 4], [N], out=out)
            lat.append((time.perf_counter() - ts) * 1e3)
            st = e.last_stats()
            ph.append((st.key_stage_ms, st.bucket_ms, st.lookup_ms, st.expand_ms, st.unique_keys, max(st.shard_ms[:P]), max(st.sent[:P]), st.unique_misses,
                       max(st.copy_wait_ms[:P]), st.copied_bytes))
        sync_all()
        dt = time.perf_counter() - t1
        ok = check(e, out, bt[(steps - 1) % 4])
        ok_all &= ok
        pm = np.mean(np.array(ph, dtype=np.float64), axis=0)
        own_dev = entry_devs[0]
        st = e.last_stats()
        remote = float(sum(st.sent[s] for s in range(P) if shard_devs[s] != own_dev))
        one = {"transport": tr, "pieces_per_shard": [int(x) for x in st.passes[:P]], "copy_wait_ms_slowest_shard": float(pm[8]),
               "row_bytes_copied_per_request": float(pm[9]),
               "lookups_per_s": N * steps / dt, "ms_per_request": dt / steps * 1e3, "p50_request_ms": float(np.percentile(lat, 50)),
               "p99_request_ms": float(np.percentile(lat, 99)), "parity": ok,
               "phase_ms": {"key_stage": float(pm[0]), "bucket": float(pm[1]), "shard_lookups": float(pm[2]), "expand_repeats": float(pm[3])},
               "host_key_bytes_over_pcie": int(st.key_bytes), "dedup_level": int(st.dedup_level), "distinct_keys_per_request": float(pm[4]), "distinct_misses_per_request": float(pm[7]), "slowest_shard_ms": float(pm[5]), "largest_bucket_keys": float(pm[6]),
               "row_bytes_from_other_gpus_per_request": remote * 4 * D,
               "rows_GBps_into_entry_gpu": remote * 4 * D / (pm[2] * 1e-3) / 1e9 if pm[2] > 0 and remote else None}
        # device keys (an ensemble step upstream holds them in HBM)
        dk = [torch.from_numpy(b).to(out.device) for b in bt[:2]]
        for i in range(2):
            e.lookup_device(dk[i % 2], [N], out=out)
        sync_all()
        t1 = time.perf_counter()
        for i in range(steps):
            e.lookup_device(dk[i % 2], [N], out=out)
        sync_all()
        one["device_keys_lookups_per_s"] = N * steps / (time.perf_counter() - t1)
        del dk
        # a request on EVERY instance at once (P entry sessions, each driving its own P shard sessions)
        bar = threading.Barrier(P + 1)
        errs = []

        def work(r):
            try:
                torch.cuda.set_device(entry_devs[r])
                for i in range(2):
                    entries[r].lookup(bt[(r + i) % 4], [N], out=outs[r])
                bar.wait()
                bar.wait()
                for i in range(steps):
                    entries[r].lookup(bt[(r + i) % 4], [N], out=outs[r])
                bar.wait()
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex)[:200])
                bar.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
        [x.start() for x in th]
        thr0 = thr1 = None
        try:
            bar.wait()
            sync_all()
            thr0 = cpu_throttle_stat()
            t1 = time.perf_counter()
            bar.wait()
            bar.wait()
            sync_all()
            dtc = time.perf_counter() - t1
            thr1 = cpu_throttle_stat()
        except threading.BrokenBarrierError:
            dtc = float("nan")
        [x.join() for x in th]
        if errs:
            one["all_instances_error"] = errs[0]
        else:
            okc = all(check(entries[r], outs[r], bt[(r + steps - 1) % 4]) for r in range(P))
            ok_all &= okc
            one["all_instances_at_once"] = {"instances": P, "lookups_per_s": P * N * steps / dtc, "ms_per_round": dtc / steps * 1e3, "parity": okc,
                                            "cpu_quota_throttled_ms": (thr1[1] - thr0[1]) / 1e3 if thr0 and thr1 else None}
        if tr == transports[0]:
            res[kind] = one
        res.setdefault("by_transport", {}).setdefault(tr, {})[kind] = one
    res["transports"] = list(transports)
    res["lookups_per_s"] = (res.get("uniform") or {}).get("lookups_per_s")
    res["parity"] = bool(ok_all)
    for e in entries:
        e.close()
    ps.close()
    return res


def c3_rccl_threads_leg(a, torch, hps, ranks, rows_total, steps=30):
    """The SPMD variant of config 3 (csrc/cache/shard_session.cpp: RCCL send/recv groups) with `ranks` RCCL ranks inside THIS
    process — one thread, one device, one communicator per rank — so that a plain `python bench.py --gpus N` drives RCCL with N
    ranks as torch.distributed.run would with N processes."""
    import ctypes as C
    from oracle import hps_oracle as O
    P, D = ranks, a.dim
    N = a.tables * a.batch
    n_local = N // P
    recv_cap = int(n_local * 1.25) + 4096
    uid = (C.c_uint8 * 128)()
    hps._check(hps.LIB.hps_shard_unique_id(uid))
    servers, sessions, shards = [None] * P, [None] * P, [None] * P
    errs = []
    t0 = time.time()

    def make(r):
        try:
            torch.cuda.set_device(r)
            model = f"c3_rccl_{r}"
            cfg = {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8},
                   "models": [{"model": model, "sparse_files": ["synthetic://shard"], "num_of_worker_buffer_in_pool": 2,
                               "embedding_vecsize_per_table": [D], "maxnum_catfeature_query_per_table_per_sample": [1],
                               "default_value_for_each_table": [0.0], "deployed_device_list": [r], "max_batch_size": recv_cap,
                               "gpucache": True, "gpucacheper": 1.0, "gpucache_load_factor": 0.6, "hit_rate_threshold": 1.0}]}
            ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
            ps.load_table_synthetic(model, 0, SEED, 0, rows_total, shard=r, num_shards=P)
            ps.create_embedding_cache_per_model(model)
            sess = hps.LookupSession.create(ps, model, ps.get_embedding_cache(model, r))
            h = C.c_void_p()
            hps._check(hps.LIB.hps_shard_session_create(sess._h, r, P, uid, n_local, C.byref(h)))   # collective: ncclCommInitRank
            servers[r], sessions[r], shards[r] = ps, sess, h
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex)[:200])

    th = [threading.Thread(target=make, args=(r,)) for r in range(P)]
    [x.start() for x in th]
    [x.join() for x in th]
    if errs:
        return {"ranks": P, "error": errs[0]}
    t_setup = time.time() - t0
    rng = np.random.default_rng(SEED + 505)
    bt = [[rng.integers(0, rows_total, n_local, dtype=np.int64) for _ in range(4)] for _ in range(P)]
    outs = [torch.empty(n_local * D, dtype=torch.float32, device=torch.device("cuda", r)) for r in range(P)]
    bar = threading.Barrier(P + 1)
    tim = [[] for _ in range(P)]
    caps = [0] * P

    def work(r):
        try:
            torch.cuda.set_device(r)
            for i in range(4):
                hps._check(hps.LIB.hps_shard_session_lookup_host(shards[r], bt[r][i % 4].ctypes.data, n_local, outs[r].data_ptr()))
            bar.wait()
            bar.wait()
            for i in range(steps):
                hps._check(hps.LIB.hps_shard_session_lookup_host(shards[r], bt[r][i % 4].ctypes.data, n_local, outs[r].data_ptr()))
                t = [C.c_float(0) for _ in range(3)]
                recv, kb = C.c_uint64(0), C.c_int32(0)
                hps._check(hps.LIB.hps_shard_session_last_timing(shards[r], C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), C.byref(recv), C.byref(kb)))
                tim[r].append([x.value for x in t])
            att, cap = C.c_uint32(0), C.c_uint64(0)
            sent = (C.c_uint64 * P)()
            hps._check(hps.LIB.hps_shard_session_last_stats(shards[r], C.byref(cap), C.byref(att), sent, P))
            caps[r] = cap.value
            bar.wait()
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex)[:200])
            bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [x.start() for x in th]
    try:
        bar.wait()
        t1 = time.perf_counter()
        bar.wait()
        bar.wait()
        for r in range(P):
            torch.cuda.synchronize(r)
        dt = time.perf_counter() - t1
    except threading.BrokenBarrierError:
        dt = float("nan")
    [x.join() for x in th]
    if errs:
        return {"ranks": P, "error": errs[0]}
    ok = True
    for r in range(P):
        kh = bt[r][(steps - 1) % 4]
        idx = np.linspace(0, n_local - 1, 128).astype(np.int64)
        exp = np.concatenate([O.c_synth_rows(SEED, 0, int(kh[i]), 1, D) for i in idx]).reshape(128, D)
        got = outs[r].view(-1, D)[torch.from_numpy(idx).to(outs[r].device)].cpu().numpy()
        ok &= bool(np.array_equal(got.view(np.uint32), exp.view(np.uint32)))
    tm = np.mean(np.array([x for r in range(P) for x in tim[r]], dtype=np.float64), axis=0)
    cap = caps[0]
    out = {"ranks": P, "ranks_are": "threads of this process, one device + one RCCL communicator each", "rows_total": rows_total,
           "lookups_per_s": N * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "parity": ok, "setup_seconds": t_setup,
           "keys_exchange_ms": float(tm[0]), "local_lookup_ms": float(tm[1]), "rows_exchange_ms": float(tm[2]),
           "block_capacity_keys": cap, "row_block_bytes_per_peer": cap * 4 * D,
           "rows_GBps_per_link": cap * 4 * D / (tm[2] * 1e-3) / 1e9 if tm[2] > 0 and P > 1 else None,
           "rows_frac_of_153_GBps_link": cap * 4 * D / (tm[2] * 1e-3) / 1e9 / 153.0 if tm[2] > 0 and P > 1 else None}
    for h in shards:
        hps.LIB.hps_shard_session_destroy(h)
    for s_ in sessions:
        s_.close()
    for p_ in servers:
        p_.close()
    return out


def sharded_leg(a, torch, dist, hps, rank, world, local_rank, shared_gpu):
    """One table of Rt rows x D sharded over the P ranks (owner = mix64(key) mod P), every rank resident at 100 % in
    its own HBM; per step every rank looks up N/P uniform keys through the engine's sharded session
    (csrc/cache/shard_session.cpp: bucket by owner into fixed-capacity blocks, RCCL send/recv group of the keys, padded
    local lookup, RCCL send/recv group of the rows, gather back — no count exchange, no host round trip inside a step).
    Global lookups/s = N x steps / time."""
    from hugectr_backend_amd.sharded import ShardedLookup
    P, D = world, a.dim
    coll_dev = "cpu"   # the process group is gloo: it carries the RCCL unique id, the barriers and the checks, nothing else
    torch.cuda.set_device(local_rank)
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
    # own GPU per rank: the engine's native RCCL session (no torch collective per step); ranks sharing the development box's
    # one GPU: the torch.distributed variant over gloo (control flow only)
    sl = ShardedLookup(sess, max_local_keys=n_local, native=not shared_gpu)
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
    lat, tim = [], []
    for i in range(steps):
        ts = time.perf_counter()
        out = sl.lookup(batches[i % 8])
        torch.cuda.current_stream().synchronize()
        lat.append((time.perf_counter() - ts) * 1e3)
        if getattr(sl, "last_timing", None):
            tim.append((sl.last_timing["keys_exchange_ms"], sl.last_timing["lookup_ms"], sl.last_timing["rows_exchange_ms"]))
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
        "ranks": P, "ranks_are": "processes (torch.distributed.run), one GPU each",
        "lookups_per_s": N * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
        "p50_step_latency_ms": float(np.percentile(lat, 50)),
        "rows_bytes_sent_per_rank_per_step": sent_remote * 4 * D,
        "row_exchange_GBps_per_rank": sent_remote * 4 * D / (dt / steps) / 1e9,
        "xgmi_note": "row all-to-all is bounded by 7 links x ~153 GB/s per GPU (MI355X_MICROARCH.md); the figure above "
                     "divides by the WHOLE step time, not the collective's own",
        "parity_vs_oracle_bit_exact": bool(okt.item()), "backend": dist.get_backend(), "setup_seconds": t_setup,
        "exchange": "native RCCL session (hps_shard_session_*)" if sl._native else "torch.distributed all_to_all (ranks share a GPU)",
        "block_capacity_keys": getattr(sl, "last_capacity", None), "attempts_last_step": sl.last_attempts,
    }
    if tim and getattr(sl, "last_capacity", None):
        # the two RCCL send/recv groups on their own (HIP events on the session's stream, this rank): every peer pair moves one
        # block over its own xGMI link, so the per-link rate is block bytes / group time
        tm = np.mean(np.array(tim), axis=0)
        cap = sl.last_capacity
        res["rccl_groups"] = {
            "keys_exchange_ms": float(tm[0]), "local_lookup_ms": float(tm[1]), "rows_exchange_ms": float(tm[2]),
            "row_block_bytes_per_peer": cap * 4 * D, "key_block_bytes_per_peer": (cap + 2) * 8,
            "rows_GBps_per_link": cap * 4 * D / (tm[2] * 1e-3) / 1e9 if tm[2] > 0 else None,
            "rows_frac_of_153_GBps_link": cap * 4 * D / (tm[2] * 1e-3) / 1e9 / 153.0 if tm[2] > 0 else None,
            "keys_GBps_per_link": (cap + 2) * 8 / (tm[0] * 1e-3) / 1e9 if tm[0] > 0 else None,
            "padding_fraction_of_row_blocks": 1.0 - (n_local / P) / cap,
        }
    sess.close()
    return res


if __name__ == "__main__":
    main()
