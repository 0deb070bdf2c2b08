import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu(patience_s: float = 0.0) -> bool:
    """hipGetDeviceCount() > 0, asked again for up to `patience_s` seconds: right after another process let go of
    the GPU the runtime can report no device for a moment (seen once on the box between two pytest runs)."""
    import time
    deadline = time.time() + patience_s
    if patience_s > 0:
        # first from throw-away processes (a runtime that initialised without a device may keep saying so), then in ours
        from hugectr_backend_amd.gpu_wait import wait_for_gpu
        wait_for_gpu(patience_s)
    while True:
        try:
            from hugectr_backend_amd import hps
            if hps.device_count() > 0:
                return True
        except Exception:
            pass
        if time.time() >= deadline:
            return False
        time.sleep(1.0)


def pytest_collection_modifyitems(config, items):
    markexpr = config.getoption("markexpr", "") or ""
    asked_for_gpu = "gpu" in markexpr and "not gpu" not in markexpr
    if _have_gpu(30.0 if asked_for_gpu else 0.0):
        return
    if asked_for_gpu:
        # `-m gpu` on a box without a usable device: run the tests and let them fail loudly rather than skip —
        # a green run that executed nothing would hide exactly the failure these tests exist to catch
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Build (if stale) the native libraries and the C oracle once per test session."""
    from hugectr_backend_amd import build as _b
    _b.build()
    from oracle import hps_oracle
    hps_oracle.build()
    # On a box with a GPU: bring torch and the ROCm libraries into the page cache HERE, where nothing times it.  The first
    # `import torch` on a fresh box takes minutes while the image pages in (2.7 x longer than usual on one box of round 5, where the
    # suite's first GPU test — a subprocess with a 300-s limit — was killed inside exactly that import).
    try:
        from hugectr_backend_amd import hps
        if hps.device_count() > 0:
            import torch
            if torch.cuda.is_available():
                torch.zeros(1, device="cuda").cpu()
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture
def plain_lru(monkeypatch):
    """Caches created inside the test take every new key (HPS_LRU_ADMIT=0): for tests that watch the insert kernel's mechanics
    (what was missed is resident afterwards) rather than the default admission rule, which keeps a key seen once out of a
    bucket whose keys were all hit lately."""
    monkeypatch.setenv("HPS_LRU_ADMIT", "0")


def make_tables(spec, seed=20260929, key_space_mult=3, rng=None):
    """spec: list of (R, D).  Keys are a random subset of [0, key_space_mult*R) in random order."""
    from oracle import hps_oracle as O
    rng = rng or np.random.default_rng(seed)
    out = []
    for t, (R, D) in enumerate(spec):
        keys = rng.permutation(R * key_space_mult)[:R].astype(np.int64)
        rows = O.np_synth_rows(seed, t, keys, D)
        out.append((keys, rows))
    return out


def ps_config(model, tables, dirs=None, gpucache=True, gpucacheper=0.5, hit_rate_threshold=1.0, defaults=None,
              maxcat=None, max_batch=1024, extra=None, device=0):
    T = len(tables)
    m = {
        "model": model,
        "sparse_files": dirs or [f"/nonexistent/{model}_{t}" for t in range(T)],
        "num_of_worker_buffer_in_pool": 3,
        "embedding_vecsize_per_table": [int(r.shape[1]) for _, r in tables],
        "maxnum_catfeature_query_per_table_per_sample": maxcat or [1] * T,
        "default_value_for_each_table": defaults or [0.0] * T,
        "deployed_device_list": [device],
        "max_batch_size": max_batch,
        "gpucache": gpucache,
    }
    if gpucache:
        m["hit_rate_threshold"] = hit_rate_threshold
        m["gpucacheper"] = gpucacheper
    if extra:
        m.update(extra)
    return {"supportlonglong": True, "volatile_db": {"type": "hash_map", "num_partitions": 8}, "models": [m]}
