"""Kernel-level timing of the config-5 dense step alone (bottom MLP + dot interaction) on random OUTPUT0-shaped
input: `python tools/dense_bench.py [batch] [tables] [dim]`, or under `rocprofv3 --kernel-trace --stats` for the
per-kernel split."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from hugectr_backend_amd.dense import DenseInteraction
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 26
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rng = np.random.default_rng(0)
    dims, k = [512, 256, D], 13
    ws, bs = [], []
    for n in dims:
        ws.append(((rng.random((k, n), dtype=np.float32) * 2 - 1) / np.sqrt(k)).astype(np.float32))
        bs.append((rng.random(n, dtype=np.float32) * 0.1).astype(np.float32))
        k = n
    op = DenseInteraction(ws, bs, T, D)
    x = torch.randn(B, 13, device="cuda")
    emb = torch.rand(T * B * D, device="cuda") - 0.5
    out = torch.empty((B, op.out_stride), dtype=torch.float16, device="cuda")
    for _ in range(5):
        op.forward(x, emb, B, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    iters = 50
    for _ in range(iters):
        op.forward(x, emb, B, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = T * B * D * 4 + B * 13 * 4 + 2 * B * D * 2 + B * op.out_stride * 2
    print(f"B={B} T={T} D={D}: {ms*1e3:.1f} us per forward, {nbytes/ms/1e6:.0f} GB/s algorithmic "
          f"({nbytes/ms/1e6/8000:.3f} of 8 TB/s), {B/ms/1e3:.1f} M samples/s")


if __name__ == "__main__":
    main()
