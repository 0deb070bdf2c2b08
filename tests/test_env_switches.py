"""The environment switches the product still reads (INTEGRATION.md §4 lists them; round 5 removed the 27 that were A/B switches
of decided experiments).  Each one is exercised here in a process of its own — the libraries read them once, at load or at the
first use — unless another test file already owns it (named in INTEGRATION.md).  Rows are compared with the oracle every time:
a switch may change how the path runs, never what it returns."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(code, env, timeout=900):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"rc={r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-2500:]}"
    return r


_HOST_LOOKUP = """
    import numpy as np
    from tests.conftest import make_tables, ps_config
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    O.build()
    tables = make_tables([(20000, 16), (5000, 8)])
    ps = hps.HierParameterServer.create_from_dict(ps_config("m", tables, gpucache=False, max_batch=8192), load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays("m", t, k, r)
    s = hps.LookupSession.create(ps, "m", None)
    def threads():
        return int([l for l in open("/proc/self/status") if l.startswith("Threads:")][0].split()[1])
    before = threads()
    rng = np.random.default_rng(0)
    nk = [8000, 3000]
    q = np.concatenate([rng.choice(tables[0][0], nk[0]), rng.choice(tables[1][0], nk[1])]).astype(np.int64)
    out = s.lookup(q, nk)
    assert np.array_equal(out.view(np.uint32), O.np_lookup(tables, q, nk, [0.0, 0.0]).view(np.uint32))
    print("THREADS", before, threads())
"""


def test_pool_sizes_follow_the_environment():
    """HCTR_DEFAULT_CONCURRENCY (the reference's variable, thread_pool.cpp:25-41 there): CPUs to assume — the general pool gets one
    worker less; HPS_SERVING_THREADS: workers of the per-request gather pool.  Both pools start lazily."""
    r = _run(_HOST_LOOKUP, {"HCTR_DEFAULT_CONCURRENCY": 3, "HPS_SERVING_THREADS": 5})
    before, after = map(int, r.stdout.split("THREADS")[1].split())
    # (table loading started the general pool: 2 workers; the first lookup starts the serving pool: 5 workers)
    assert after - before == 5, (before, after)
    r = _run(_HOST_LOOKUP, {"HCTR_DEFAULT_CONCURRENCY": 6, "HPS_SERVING_THREADS": 2})
    before, after = map(int, r.stdout.split("THREADS")[1].split())
    assert after - before == 2, (before, after)


_POOL_THREADS = """
    import os
    def pool_threads():
        out = {}
        for tid in os.listdir("/proc/self/task"):
            try:
                name = open(f"/proc/self/task/{tid}/comm").read().strip()
                if name in ("hps-pool", "hps-serving"):
                    allowed = [l.split(":")[1].strip() for l in open(f"/proc/self/task/{tid}/status") if l.startswith("Cpus_allowed_list")][0]
                    out.setdefault(name, set()).add(allowed)
            except OSError:
                pass
        return out
    def cpus(spec):
        s = set()
        for part in spec.split(","):
            a, _, b = part.partition("-")
            s.update(range(int(a), int(b or a) + 1))
        return s
    mine = cpus([l.split(":")[1].strip() for l in open("/proc/self/status") if l.startswith("Cpus_allowed_list")][0])
"""


def test_worker_pools_can_be_bound_to_a_numa_node_or_left_alone():
    """HPS_NUMA_NODE=<n>: the workers of both pools run on the CPUs of that node (and the tables they load are first touched
    there); HPS_NUMA_NODE=off: wherever the scheduler puts them.  Unset: the first server of the process decides (the node of the
    deployed GPUs when they share one, none without GPU caches or on a one-node machine) — the GPU test below."""
    code = _HOST_LOOKUP + _POOL_THREADS + """
    node0 = cpus(open("/sys/devices/system/node/node0/cpulist").read().strip()) & mine
    th = pool_threads()
    assert set(th) == {"hps-pool", "hps-serving"}, th
    print("NODE", hps.pool_numa_node())
    for name, lists in th.items():
        for l in lists:
            assert cpus(l) == (node0 if BOUND else mine), (name, l, sorted(node0)[:4], BOUND)
    """
    r = _run("BOUND = True\n" + textwrap.dedent(code), {"HPS_NUMA_NODE": 0, "HPS_SERVING_THREADS": 3, "HCTR_DEFAULT_CONCURRENCY": 4})
    assert "NODE 0" in r.stdout
    r = _run("BOUND = False\n" + textwrap.dedent(code), {"HPS_NUMA_NODE": "off", "HPS_SERVING_THREADS": 3, "HCTR_DEFAULT_CONCURRENCY": 4})
    assert "NODE -1" in r.stdout


def test_roctx_ranges_can_be_switched_on_without_a_profiler():
    """HPS_ENABLE_ROCTX=1: roctx ranges around the plugin's phases (the reference's NVTX ranges, hps.cc:375,671).  With no profiler
    attached — and on a box without the library — the requests are served all the same."""
    code = """
        import json, numpy as np, tempfile, pathlib
        from tests import triton_mock as tm
        from tests.conftest import make_tables, ps_config
        from oracle import hps_oracle as O
        O.build()
        tables = make_tables([(3000, 8)])
        d = pathlib.Path(tempfile.mkdtemp())
        O.np_write_table(d / "t0", *tables[0])
        cfg = ps_config("m", tables, dirs=[str(d / "t0")], gpucache=False, max_batch=1024)
        (d / "ps.json").write_text(json.dumps(cfg))
        srv = tm.Server(d / "ps.json")
        inst = srv.load_model("m", tm.model_config("m", kind="KIND_CPU", gpus=[])).create_instance("m_0", tm.KIND_CPU, 0)
        q = np.random.default_rng(1).choice(tables[0][0], 500).astype(np.int64)
        req = tm.Request("1")
        req.add_input("KEYS", q.reshape(1, -1)).add_input("NUMKEYS", np.asarray([[500]], np.int32)).request_output("OUTPUT0")
        inst.execute([req])
        assert req.error_code == -1, req.error_message
        assert np.array_equal(req.output_numpy().view(np.uint32), O.np_lookup(tables, q, [500], [0.0]).view(np.uint32))
        srv.shutdown()
        print("OK")
    """
    assert "OK" in _run(code, {"HPS_ENABLE_ROCTX": 1}).stdout


_GPU_LOOKUP = """
    import numpy as np
    from tests.conftest import make_tables, ps_config
    from hugectr_backend_amd import hps
    from oracle import hps_oracle as O
    O.build()
    T = 3
    tables = make_tables([(60000, 128)] * T)
    cfg = ps_config("m", tables, gpucacheper=0.3, max_batch=60000, maxcat=[1] * T, extra=EXTRA)
    ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
    for t, (k, r) in enumerate(tables):
        ps.load_table_arrays("m", t, k, r)
    ps.create_embedding_cache_per_model("m")
    s = hps.LookupSession.create(ps, "m", ps.get_embedding_cache("m", 0))
    s.set_option("timing", 1)
    rng = np.random.default_rng(3)
    for nk in ([50000, 50000, 50000], [3000, 1, 0]):      # a big request (staged in pieces) and a small one
        q = np.concatenate([rng.choice(tables[t][0], nk[t]) for t in range(T)]).astype(np.int64)
        q[::13] = -9 - np.arange(q[::13].size)
        out = s.lookup(q, nk).cpu().numpy()
        assert np.array_equal(out.view(np.uint32), O.np_lookup(tables, q, nk, [0.0] * T).view(np.uint32)), nk
    st = s.last_stats()
    print("STATS", st.probe_gather_ms, st.phase_ms[3])
"""


@pytest.mark.gpu
def test_worker_pools_follow_the_gpu_to_its_numa_node():
    """No HPS_NUMA_NODE: on a machine with several NUMA nodes the pools of a one-GPU deployment are bound to the node the GPU hangs
    off; on a one-node machine they are not bound at all.  Rows exact either way."""
    code = "EXTRA = {}\n" + textwrap.dedent(_GPU_LOOKUP) + textwrap.dedent(_POOL_THREADS) + textwrap.dedent("""
        import ctypes as C, glob
        hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        b = C.create_string_buffer(64)
        assert hip.hipDeviceGetPCIBusId(b, 64, 0) == 0
        gpu_node = int(open(f"/sys/bus/pci/devices/{b.value.decode().lower()}/numa_node").read())
        nodes = len(glob.glob("/sys/devices/system/node/node[0-9]*"))
        node = hps.pool_numa_node()
        if nodes > 1 and gpu_node >= 0:
            assert node == gpu_node, (node, gpu_node)
            want = cpus(open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()) & mine
            for name, lists in pool_threads().items():
                assert all(cpus(l) == want for l in lists), (name, lists)
        else:
            assert node == -1
        import threading
        res = {}
        def t():
            res["first"] = hps.bind_calling_thread(); res["aff"] = os.sched_getaffinity(0); res["second"] = hps.bind_calling_thread()
        th = threading.Thread(target=t); th.start(); th.join()
        # (a process that was started inside one node already — numactl, or a parent whose thread the plugin had placed — stays as it is)
        inside_one = any(mine <= cpus(open(p).read().strip()) for p in glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"))
        if nodes > 1 and gpu_node >= 0 and not inside_one:
            assert res["first"] and set(res["aff"]) == want and not res["second"], res
        else:
            assert not res["first"]
        print("POOLS ON NODE", node, "OF", nodes)
    """)
    assert "POOLS ON NODE" in _run(code, {}).stdout


@pytest.mark.gpu
def test_slow_call_trace_prints_where_a_call_spent_its_time():
    """HPS_TRACE_TAIL=<ms>: every lookup slower than that writes one line to stderr (key staging, enqueues, count read-back, host
    gather, final synchronisation) — the diagnostic behind DESIGN.md §4's stall hunts."""
    r = _run("EXTRA = {}\n" + textwrap.dedent(_GPU_LOOKUP), {"HPS_TRACE_TAIL": "0.0001"})
    assert "[hps call]" in r.stderr and "key staging" in r.stderr
    r = _run("EXTRA = {}\n" + textwrap.dedent(_GPU_LOOKUP), {})
    assert "[hps call]" not in r.stderr


@pytest.mark.gpu
def test_copy_engine_wakeup_can_be_switched_off_and_called_by_hand():
    """HPS_WAKE_COPY_ENGINES=0: cache creation does not touch the SDMA engines (the sanitizer jobs run that way); the explicit call
    still works and reports the engines it woke."""
    code = "EXTRA = {}\n" + textwrap.dedent(_GPU_LOOKUP) + textwrap.dedent("""
        n, report = hps.wake_copy_engines(0)
        assert n >= 1 and "engine" in report, (n, report)
        print("WOKE", n)
    """)
    assert "WOKE" in _run(code, {"HPS_WAKE_COPY_ENGINES": 0}).stdout


@pytest.mark.gpu
@pytest.mark.parametrize("interleave", [0, 1])
def test_page_locked_host_tier_with_and_without_numa_interleaving(interleave):
    """HPS_HOST_NUMA_INTERLEAVE: the page-locked host tier of a ps_direct_access model interleaved over the NUMA nodes (default: on
    when the machine has several GPUs and several nodes) or left to HIP's placement.  Same rows either way."""
    r = _run('EXTRA = {"ps_direct_access": True}\n' + textwrap.dedent(_GPU_LOOKUP), {"HPS_HOST_NUMA_INTERLEAVE": interleave})
    assert "STATS" in r.stdout
