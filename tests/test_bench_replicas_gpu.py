"""bench.py --gpus N: the reference's multi-GPU arrangement (docs/architecture.md:11,29; model_state.cpp:395-420) — ONE process,
ONE parameter server, one embedding cache per deployed device.  On the 1-GPU box the replicas share device 0 (the second one
deploys the model again under its own name), which drives the same multi-cache code path: two caches built by one server,
their sessions driven side by side by threads of one process, every replica's rows checked."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SMALL = ["--rows", "300000", "--batch", "4096", "--steps", "3", "--warmup", "1", "--blocks", "2",
         "--no-cpu-baseline", "--no-extra-legs", "--sharded-steps", "4", "--shard-rows", "400000"]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_replicas_in_one_process_share_one_parameter_server():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *SMALL], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert "reduced" not in d["config"]["workload"] and "300000 rows/table" in d["config"]["workload"]
    assert "ONE process, ONE parameter server" in d["config"]["parallelism"]
    assert len(d["per_gpu"]) == 2 and {p["model"] for p in d["per_gpu"]} == {"criteo_dlrm", "criteo_dlrm_dup1"}
    assert d["parity_vs_oracle_bit_exact"] is True and d["parity_full_batch_vs_direct_row_index"] is True
    assert d["per_gpu"][1]["parity_full_batch_vs_direct_row_index"] is True
    # whole-job value = both replicas' keys over the block time
    assert abs(d["value"] - 2 * 3 * 26 * 4096 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_driver_launch_with_two_ranks_rank0_serves_every_gpu():
    """The driver's N > 1 command line: rank 0 runs the replicas measurement for both GPUs, rank 1 waits and joins the
    sharded-table leg (config 3), here over gloo because the two ranks share the one GPU."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(ROOT / "bench.py"), "--gpus", "2", *SMALL],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and len(d["per_gpu"]) == 2
    assert d["parity_full_batch_vs_direct_row_index"] is True
    leg = d["extra_legs"]["sharded_c3"]
    assert "error" not in leg, leg
    assert leg["parity_vs_oracle_bit_exact"] is True
