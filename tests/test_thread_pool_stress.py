"""Fork-join pool of the host tier (csrc/ps/thread_pool.cpp) under many concurrent callers.

Round 5 found a slot-reuse race in the lock-free fast path: a worker still looking at a slot's finished loop could claim — and run —
tasks of the NEXT loop published in that slot before its claim word was there; tasks ran twice and the owner's wait for
`done == n` never ended (one hang in ~60 bench runs; one in two runs of the sharded stress driver, where 9 lookup sessions and 3
entry sessions fork-join side by side).  tools/micro/forkjoin_stress.cpp — 14 callers on 13 workers, loops of 2-61 tasks, every
task index summed — reproduced it within two seconds before the fix."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_many_concurrent_fork_joins_run_every_task_exactly_once(tmp_path):
    exe = tmp_path / "forkjoin_stress"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", f"-I{ROOT / 'hugectr_backend_amd' / 'csrc'}",
                    str(ROOT / "tools" / "micro" / "forkjoin_stress.cpp"), str(ROOT / "hugectr_backend_amd" / "csrc" / "ps" / "thread_pool.cpp"),
                    "-o", str(exe)], check=True, timeout=300)
    for callers in (14, 5):
        r = subprocess.run([str(exe), str(callers), "6"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (callers, r.stdout[-400:], r.stderr[-400:])
        assert "WRONG SUM" not in r.stderr and "HANG" not in r.stdout
