"""Host-side handle of the dense step of BASELINE config 5 (DLRM bottom MLP + dot interaction) — a thin ctypes
wrapper over the hps_dense_* entry points of libhps_amd.so (include/hps_amd.h).  The arithmetic is the HIP/MFMA
code in csrc/dense/; torch is used for device buffers only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import hps


class DenseInteraction:
    """weights[l]: fp32 [K_l, dims[l]] (x @ W convention), biases[l]: fp32 [dims[l]]."""

    def __init__(self, weights, biases, num_tables: int, emb_dim: int, device: int = 0):
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        if not ws or len(ws) != len(bs):
            raise hps.HpsError(hps.ERR_INVALID_ARG, "one weight matrix and one bias vector per layer")
        num_dense = ws[0].shape[0]
        k = num_dense
        for w, b in zip(ws, bs):
            if w.ndim != 2 or w.shape[0] != k or b.shape != (w.shape[1],):
                raise hps.HpsError(hps.ERR_INVALID_ARG, "layer shapes do not chain")
            k = w.shape[1]
        L = len(ws)
        dims = (C.c_uint32 * L)(*[w.shape[1] for w in ws])
        wp = (C.c_void_p * L)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * L)(*[b.ctypes.data for b in bs])
        h = C.c_void_p()
        hps._check(hps.LIB.hps_dense_create(device, num_dense, L, dims, wp, bp, num_tables, emb_dim, C.byref(h)))
        self._h = h
        self.device = device
        self.num_dense = num_dense
        self.num_tables = num_tables
        self.emb_dim = emb_dim
        self.out_dim = int(hps.LIB.hps_dense_out_dim(h))
        self.out_stride = int(hps.LIB.hps_dense_out_stride(h))

    def forward(self, dense, embeddings, batch: int, out=None):
        """dense: CUDA fp32 [batch, num_dense]; embeddings: CUDA fp32, the lookup's OUTPUT0 (table-major
        [num_tables, batch, emb_dim], flat is fine).  Returns a CUDA fp16 tensor [batch, out_stride] (columns
        [0, out_dim) are the result, the rest zero padding).  Runs on torch's current stream."""
        import torch
        assert dense.is_cuda and embeddings.is_cuda and dense.dtype == torch.float32 and embeddings.dtype == torch.float32
        assert dense.is_contiguous() and embeddings.is_contiguous()
        assert dense.numel() == batch * self.num_dense and embeddings.numel() == self.num_tables * batch * self.emb_dim
        if out is None:
            out = torch.empty((batch, self.out_stride), dtype=torch.float16, device=dense.device)
        stream = torch.cuda.current_stream(dense.device).cuda_stream
        hps._check(hps.LIB.hps_dense_forward(self._h, dense.data_ptr(), embeddings.data_ptr(), batch, out.data_ptr(),
                                             C.c_void_p(stream)))
        return out

    def lookup_interact(self, session, keys, batch: int, dense, out=None):
        """Fused arrangement: `keys` CUDA int64 [num_tables * batch] (table-major, one key per table per sample)
        looked up through `session` (a LookupSession of a ps_direct_access model) and fed to the interaction from
        the cache slots / miss staging directly — no OUTPUT0.  Returns the same tensor as forward().  Blocking."""
        import torch
        assert keys.is_cuda and keys.dtype == torch.int64 and keys.is_contiguous() and keys.numel() == self.num_tables * batch
        assert dense.is_cuda and dense.dtype == torch.float32 and dense.is_contiguous() and dense.numel() == batch * self.num_dense
        if out is None:
            out = torch.empty((batch, self.out_stride), dtype=torch.float16, device=keys.device)
        torch.cuda.current_stream(keys.device).synchronize()      # the session works on its own stream
        hps._check(hps.LIB.hps_session_lookup_interact_device(session._h, self._h, keys.data_ptr(), batch, dense.data_ptr(),
                                                              out.data_ptr()))
        return out

    def close(self):
        if getattr(self, "_h", None):
            hps.LIB.hps_dense_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
