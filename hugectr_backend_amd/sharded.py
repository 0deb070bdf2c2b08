"""Table-sharded (model-parallel) lookup — BASELINE config 3.  Thin binding.

Not in the reference (which only replicates: one independent cache per GPU); this is the
north-star addition: one big table whose rows are partitioned over the P ranks of a node,
``owner(key) = mix64(key) mod P``.

GPU-cache sessions on a real multi-GPU node (process group backend "nccl"): the whole exchange runs inside the engine
(`hps_shard_session_*`, csrc/cache/shard_session.cpp): RCCL send/recv groups on the lookup session's stream with
fixed-capacity blocks — no count exchange, no host read-back, no stream synchronisation inside a call.  torch.distributed
is used once, to hand rank 0's RCCL unique id to the other ranks.

The torch.distributed variant below remains for host-tier shards (gpucache=false sessions, any CPU backend such as
gloo: the world_size-2 CPU test) and for ranks that share one GPU on a development box.  Per lookup and rank it does:

    1. bucket the local keys by owner            HIP: hps_shard_bucket_device (stable counting sort)   [host tier: numpy]
    2. all-to-all of the per-destination counts  torch.distributed (RCCL over xGMI with backend "nccl")
    3. all-to-all(v) of the keys                 8 B/key
    4. local lookup of the received keys         LookupSession (GPU cache + host tier of THIS rank's shard)
    5. all-to-all(v) of the rows back            4*D B/key — the step that bounds config 3 (xGMI per-link bandwidth)
    6. restore the input order                   HIP: hps_shard_unpermute_device                        [host tier: numpy]

One process per GPU; no tracing compiler, no collective besides the two data all-to-alls and the count exchange.
With backend "gloo" and gpucache=false sessions the same code runs on CPU tensors (used by the world_size-2 tests).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import hps


def owner_of(keys, num_shards: int) -> np.ndarray:
    """mix64(key) mod P on the host (NumPy), identical to the device function in shard_kernels.hip."""
    x = np.asarray(keys, dtype=np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x % np.uint64(num_shards)).astype(np.int64)


def shard_rows(keys, rows, rank: int, num_shards: int):
    """The (keys, rows) subset of a table that `rank` owns."""
    m = owner_of(keys, num_shards) == rank
    return np.ascontiguousarray(np.asarray(keys)[m]), np.ascontiguousarray(np.asarray(rows)[m])


class ShardedLookup:
    """One sharded single-table lookup endpoint per rank.

    session : hps.LookupSession of a ONE-table model holding this rank's shard of the table
    group   : torch.distributed process group (None = default group)
    """

    def __init__(self, session: hps.LookupSession, group=None, max_local_keys: int | None = None, native: bool | None = None):
        import torch.distributed as dist
        if session.num_tables != 1:
            raise hps.HpsError(hps.ERR_UNSUPPORTED, "ShardedLookup handles one table per session")
        self.sess = session
        self.dim = session.dims[0]
        self.group = group
        self.dist = dist
        self.P = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device_mode = session.use_gpu_cache
        self.last_sent = None  # per-destination key counts of the last call (for bandwidth accounting)
        self.last_attempts = 1
        self._native = None
        if native is None:
            native = self.device_mode and dist.get_backend(group) == "nccl"
        if native:
            self._init_native(max_local_keys)

    def _init_native(self, max_local_keys):
        """Rank 0 draws the RCCL unique id, torch.distributed carries it to the others, the engine does the rest."""
        import torch
        dist = self.dist
        on_dev = dist.get_backend(self.group) == "nccl"
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            hps._check(hps.LIB.hps_shard_unique_id(buf))
            uid = torch.tensor(list(buf), dtype=torch.uint8)
        if on_dev:
            uid = uid.cuda()
        dist.broadcast(uid, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        raw = (C.c_uint8 * 128)(*uid.cpu().tolist())
        if max_local_keys is None:
            raise hps.HpsError(hps.ERR_INVALID_ARG, "the native sharded session needs max_local_keys")
        h = C.c_void_p()
        hps._check(hps.LIB.hps_shard_session_create(self.sess._h, self.rank, self.P, raw, int(max_local_keys), C.byref(h)))
        self._native = h
        self.max_local_keys = int(max_local_keys)

    def _lookup_native(self, d_keys, out=None):
        import torch
        n = d_keys.numel()
        assert d_keys.is_cuda and d_keys.dtype == torch.int64 and d_keys.is_contiguous()
        if out is None:
            out = torch.empty(max(n, 1) * self.dim, dtype=torch.float32, device=d_keys.device)
        torch.cuda.current_stream(d_keys.device).synchronize()   # the engine works on the session's own stream
        hps._check(hps.LIB.hps_shard_session_lookup(self._native, d_keys.data_ptr(), n, out.data_ptr()))
        return self._native_stats(out, n)

    def lookup_host(self, h_keys: np.ndarray, out):
        """Keys in host memory (the reference's lookup contract): staged and, when they fit, narrowed by the engine."""
        h_keys = np.ascontiguousarray(h_keys, dtype=np.int64)
        hps._check(hps.LIB.hps_shard_session_lookup_host(self._native, h_keys.ctypes.data, h_keys.size, out.data_ptr()))
        return self._native_stats(out, h_keys.size)

    def _native_stats(self, out, n):
        cap, att = C.c_uint64(0), C.c_uint32(0)
        sent = (C.c_uint64 * self.P)()
        hps._check(hps.LIB.hps_shard_session_last_stats(self._native, C.byref(cap), C.byref(att), sent, self.P))
        self.last_sent = [int(x) for x in sent]
        self.last_attempts = int(att.value)
        self.last_capacity = int(cap.value)
        t = [C.c_float(0), C.c_float(0), C.c_float(0)]
        recv, kb = C.c_uint64(0), C.c_int32(0)
        hps._check(hps.LIB.hps_shard_session_last_timing(self._native, C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), C.byref(recv), C.byref(kb)))
        self.last_timing = {"keys_exchange_ms": t[0].value, "lookup_ms": t[1].value, "rows_exchange_ms": t[2].value,
                            "keys_received": int(recv.value), "key_bytes": int(kb.value)}
        return out[: n * self.dim]

    def close(self):
        if self._native:
            hps.LIB.hps_shard_session_destroy(self._native)
            self._native = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        """all_to_all_single; with a CPU-only backend (gloo: the 1-GPU development box, where the ranks share
        a device) device tensors take a detour through host memory — the production path is RCCL on the device."""
        import torch
        if out.is_cuda and self.dist.get_backend(self.group) == "gloo":
            o = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
            out.copy_(o)
        else:
            self.dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    # ---- device path (GPU cache sessions, RCCL) -------------------------------------------------------------
    def _lookup_device(self, d_keys):
        import torch
        dist, P, D = self.dist, self.P, self.dim
        n = d_keys.numel()
        dev = d_keys.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        keys_sorted = torch.empty(n, dtype=torch.int64, device=dev)
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        totals = torch.empty(P, dtype=torch.int64, device=dev)  # written as uint64 by the kernel
        ws = torch.empty(max(int(hps.LIB.hps_shard_bucket_workspace_bytes(n, P)), 16), dtype=torch.uint8, device=dev)
        hps._check(hps.LIB.hps_shard_bucket_device(d_keys.data_ptr(), n, P, keys_sorted.data_ptr(), perm.data_ptr(),
                                                   totals.data_ptr(), ws.data_ptr(), C.c_void_p(stream)))
        # counts: what I send to each rank -> what I receive from each rank
        recv_counts = torch.empty_like(totals)
        self._a2a(recv_counts, totals)
        both = torch.stack([totals, recv_counts]).cpu().tolist()  # one device round trip for both count vectors
        send, recv = both[0], both[1]
        self.last_sent = send
        n_recv = int(sum(recv))
        keys_in = torch.empty(n_recv, dtype=torch.int64, device=dev)
        self._a2a(keys_in, keys_sorted, recv, send)
        rows_local = torch.empty(max(n_recv, 1) * D, dtype=torch.float32, device=dev)
        torch.cuda.current_stream(dev).synchronize()  # the session runs on its own stream
        if n_recv:
            self.sess.lookup_device(keys_in, [n_recv], out=rows_local)
        rows_back = torch.empty(max(n, 1) * D, dtype=torch.float32, device=dev)
        self._a2a(rows_back[: n * D], rows_local[: n_recv * D], [c * D for c in send], [c * D for c in recv])
        out = torch.empty(n * D, dtype=torch.float32, device=dev)
        hps._check(hps.LIB.hps_shard_unpermute_device(rows_back.data_ptr(), perm.data_ptr(), n, D, out.data_ptr(),
                                                      C.c_void_p(stream)))
        return out

    # ---- host path (gpucache=false sessions; any backend that moves CPU tensors, e.g. gloo) ---------------
    def _lookup_host(self, keys: np.ndarray):
        import torch
        dist, P, D = self.dist, self.P, self.dim
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        own = owner_of(keys, P)
        perm = np.argsort(own, kind="stable")
        keys_sorted = keys[perm]
        send = np.bincount(own, minlength=P).astype(np.int64)
        recv_t = torch.empty(P, dtype=torch.int64)
        dist.all_to_all_single(recv_t, torch.from_numpy(send.copy()), group=self.group)
        recv = recv_t.tolist()
        send_l = send.tolist()
        self.last_sent = send_l
        n_recv = int(sum(recv))
        keys_in = torch.empty(n_recv, dtype=torch.int64)
        dist.all_to_all_single(keys_in, torch.from_numpy(keys_sorted.copy()), output_split_sizes=recv,
                               input_split_sizes=send_l, group=self.group)
        rows_local = np.empty(n_recv * D, dtype=np.float32)
        if n_recv:
            self.sess.lookup(keys_in.numpy(), [n_recv], out=rows_local)
        rows_back = torch.empty(keys.size * D, dtype=torch.float32)
        dist.all_to_all_single(rows_back, torch.from_numpy(rows_local), output_split_sizes=[c * D for c in send_l],
                               input_split_sizes=[c * D for c in recv], group=self.group)
        out = np.empty((keys.size, D), dtype=np.float32)
        out[perm] = rows_back.numpy().reshape(keys.size, D)
        return out.ravel()

    def lookup(self, keys):
        """keys: this rank's keys (torch CUDA int64 tensor in device mode, array-like in host mode).
        Returns the rows in input order (flat fp32: torch CUDA tensor / numpy array)."""
        if self._native:
            return self._lookup_native(keys)
        if self.device_mode:
            return self._lookup_device(keys)
        return self._lookup_host(np.asarray(keys))
