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


def _line(out, cwd):
    """The compact line the driver parses (the LAST line of stdout, < 4 KB) merged over the full result bench.py leaves in
    bench_extra.json of its working directory."""
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.strip().splitlines()[-1] == lines[0], out[-2000:]
    assert len(lines[0]) < 4096
    compact = json.loads(lines[0])
    full = json.loads((Path(cwd) / "bench_extra.json").read_text())
    for k in ("value", "ms_per_step", "n_gpus", "scaling"):
        assert compact[k] == full[k], k
    assert compact["roofline"]["frac"] is not None
    return full


@pytest.mark.gpu
def test_two_replicas_in_one_process_share_one_parameter_server(tmp_path):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *SMALL], capture_output=True, text=True, timeout=600,
                       cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout, tmp_path)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert "reduced" not in d["config"]["workload"] and "300000 rows/table" in d["config"]["workload"]
    assert "ONE process, ONE parameter server" in d["config"]["parallelism"]
    assert len(d["per_gpu"]) == 2 and {p["model"] for p in d["per_gpu"]} == {"criteo_dlrm", "criteo_dlrm_dup1"}
    assert d["parity_vs_oracle_bit_exact"] is True and d["parity_full_batch_vs_direct_row_index"] is True
    assert d["per_gpu"][1]["parity_full_batch_vs_direct_row_index"] is True
    # whole-job value = both replicas' keys over the block time
    assert abs(d["value"] - 2 * 3 * 26 * 4096 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_plain_gpus_2_also_measures_both_config3_variants_and_the_cpu_baseline(tmp_path):
    """`python bench.py --gpus N` without a launcher (the form of the driver's 1-GPU command): after the replicas measurement the
    line carries BOTH config-3 variants — the table-sharded model served by entry instances (csrc/cache/shard_entry.h) and the SPMD
    session over RCCL with one rank per device as threads of the process — and the cpu_baseline (round 4's N > 1 line had neither).
    On the 1-GPU box: two logical shards on device 0, one RCCL rank."""
    small = [x for x in SMALL if x not in ("--no-extra-legs", "--no-cpu-baseline")] + ["--cpu-seconds", "1"]
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *small], capture_output=True, text=True, timeout=900,
                       cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout, tmp_path)
    compact = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    legs = compact["legs"]
    assert "c3_single_entry_P" in legs, (legs, r.stderr[-2500:])
    assert legs["c3_single_entry_P"] == 2 and legs["c3_single_entry_parity"] is True and legs["c3_single_entry_Glps"] > 0
    # both transports of the rows on the same requests (csrc/cache/shard_entry.h), and the first-contact self-test in front of them
    for tag in ("store", "copy"):
        assert legs[f"c3_single_entry_{tag}_Glps"] > 0 and legs[f"c3_single_entry_{tag}_all_instances_Glps"] > 0, tag
    st = compact["multi_gpu_selftest"]
    assert st["devices"] == [0] and st["timeout"] is False and st["error"] is None and st["rccl_allreduce_ok"] is True, st
    assert "xgmi_pair_GBps_min" in compact and "xgmi_pair_GBps_median" in compact       # (no pair on a one-GPU box: nulls)
    assert legs["c3_transport_choice"].startswith("none: every shard sat on ONE device")     # tools/choose_transport.py: nothing to choose from here
    bt = d["extra_legs"]["sharded_c3_single_entry"]["by_transport"]
    assert bt["staged_copy"]["uniform"]["parity"] and bt["staged_copy"]["uniform"]["row_bytes_copied_per_request"] > 0
    assert bt["peer_store"]["uniform"]["row_bytes_copied_per_request"] == 0
    for k in ("value_mean", "value_min", "value_max", "slow_blocks"):
        assert compact[k] is not None, k
    assert compact["config"]["measured_hit_rate"] > 0 and compact["config"]["key_bytes_over_pcie"] in (3.0, 4.0, 8.0)
    assert legs["c3_rccl_ranks"] == 1 and legs["c3_rccl_parity"] is True and legs["c3_rccl_Glps"] > 0
    assert compact["cpu_baseline"]["value"] > 0 and compact["cpu_baseline"]["kind"] == "port"
    e = d["extra_legs"]["sharded_c3_single_entry"]
    assert e["uniform"]["parity"] and e["zipf"]["parity"] and e["zipf"]["distinct_keys_per_request"] < e["uniform"]["distinct_keys_per_request"]


@pytest.mark.gpu
def test_driver_launch_with_two_ranks_rank0_serves_every_gpu(tmp_path):
    """The driver's N > 1 command line: rank 0 runs the replicas measurement for both GPUs, rank 1 waits and joins the
    sharded-table leg (config 3), here over gloo because the two ranks share the one GPU."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(ROOT / "bench.py"), "--gpus", "2", *SMALL],
                       capture_output=True, text=True, timeout=900, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout, tmp_path)
    assert d["n_gpus"] == 2 and len(d["per_gpu"]) == 2
    assert d["parity_full_batch_vs_direct_row_index"] is True
    leg = d["extra_legs"]["sharded_c3"]
    assert "error" not in leg, leg
    assert leg["parity_vs_oracle_bit_exact"] is True
