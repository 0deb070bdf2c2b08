"""The scenario of tests/test_env_switches.py::test_slow_call_trace_prints_where_a_call_spent_its_time, which hung once in the round-5
profile run: a fresh process, one session, a request of 150,000 host keys as its first lookup (tools/hang_repro_py.sh)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from tests.conftest import make_tables, ps_config  # noqa: E402
from hugectr_backend_amd import hps  # noqa: E402

T = 3
tables = make_tables([(60000, 128)] * T)
cfg = ps_config("m", tables, gpucacheper=0.3, max_batch=60000, maxcat=[1] * T, extra={})
ps = hps.HierParameterServer.create_from_dict(cfg, load_tables=False)
for t, (k, r) in enumerate(tables):
    ps.load_table_arrays("m", t, k, r)
ps.create_embedding_cache_per_model("m")
s = hps.LookupSession.create(ps, "m", ps.get_embedding_cache("m", 0))
s.set_option("timing", 1)
rng = np.random.default_rng(3)
for rep in range(6):
    for nk in ([50000, 50000, 50000], [3000, 1, 0]):
        q = np.concatenate([rng.choice(tables[t][0], nk[t]) for t in range(T)]).astype(np.int64)
        q[::13] = -9 - np.arange(q[::13].size)
        out = s.lookup(q, nk)
print("SNIPPET OK")
