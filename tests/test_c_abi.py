"""The C-ABI libraries load and export every symbol their headers declare (no compute: runs without a GPU)."""
import ctypes
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def _declared(header, prefix):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(" + prefix + r"[A-Za-z0-9_]+)\s*\(", text))


def test_libhps_amd_exports_everything_hps_amd_h_declares():
    from hugectr_backend_amd import hps
    declared = _declared("hps_amd.h", "hps_")
    assert len(declared) >= 30
    exported = _exported(ROOT / "hugectr_backend_amd" / "lib" / "libhps_amd.so")
    assert declared <= exported, sorted(declared - exported)
    assert set(hps.EXPORTED_SYMBOLS) == declared
    L = ctypes.CDLL(str(ROOT / "hugectr_backend_amd" / "lib" / "libhps_amd.so"))
    for s in declared:
        getattr(L, s)


def test_libtriton_hps_exports_the_seven_and_imports_only_declared_triton_symbols():
    lib = ROOT / "hugectr_backend_amd" / "lib" / "libtriton_hps.so"
    declared = _declared("tritonbackend_hps.h", "TRITON(?:BACKEND|SERVER)_")
    exports = {s for s in declared if s in {"TRITONBACKEND_Initialize", "TRITONBACKEND_Finalize", "TRITONBACKEND_ModelInitialize",
                                           "TRITONBACKEND_ModelFinalize", "TRITONBACKEND_ModelInstanceInitialize",
                                           "TRITONBACKEND_ModelInstanceFinalize", "TRITONBACKEND_ModelInstanceExecute"}}
    assert len(exports) == 7 and _exported(lib) == exports
    out = subprocess.run(["nm", "-D", "--undefined-only", str(lib)], capture_output=True, text=True, check=True).stdout
    imported = {l.split()[-1] for l in out.splitlines() if "TRITON" in l}
    assert imported and imported <= (declared - exports), sorted(imported - declared)
    # every import is provided by the mock core used in the tests (and by a real tritonserver)
    assert imported <= _exported(ROOT / "hugectr_backend_amd" / "lib" / "libtriton_mock_core.so")


def test_engine_fails_loudly_without_a_gpu_for_cache_models():
    import numpy as np
    import pytest
    from hugectr_backend_amd import hps
    from tests.conftest import ps_config
    if hps.device_count() > 0:
        pytest.skip("a HIP device is visible")
    tabs = [(np.arange(4, dtype=np.int64), np.zeros((4, 4), np.float32))]
    ps = hps.HierParameterServer.create_from_dict(ps_config("m", tabs, gpucache=True), load_tables=False)
    ps.load_table_arrays("m", 0, *tabs[0])
    with pytest.raises(hps.HpsError) as e:
        ps.create_embedding_cache_per_model("m")
    assert e.value.code == hps.ERR_UNAVAILABLE and "no CPU fallback" in e.value.msg


def test_dense_step_and_direct_tier_fail_loudly_without_a_gpu():
    import numpy as np
    import pytest
    from hugectr_backend_amd import hps
    from hugectr_backend_amd.dense import DenseInteraction
    from tests.conftest import ps_config
    if hps.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(hps.HpsError) as e:
        DenseInteraction([np.zeros((13, 128), np.float32)], [np.zeros(128, np.float32)], 26, 128)
    assert e.value.code == hps.ERR_UNAVAILABLE and "no CPU fallback" in e.value.msg
    tabs = [(np.arange(4, dtype=np.int64), np.zeros((4, 4), np.float32))]
    with pytest.raises(hps.HpsError) as e:
        hps.HierParameterServer.create_from_dict(ps_config("m", tabs, gpucache=True, extra={"ps_direct_access": True}),
                                                 load_tables=False)
    assert e.value.code == hps.ERR_UNAVAILABLE and "ps_direct_access needs a GPU" in e.value.msg


def test_bench_and_entry_scripts_parse_and_show_help():
    """bench.py / __graft_entry__.py are what the driver runs: they must at least parse, and bench.py must accept the
    contract's flags (--gpus/--steps/--warmup) without touching a GPU."""
    import ast
    import sys
    for f in ("bench.py", "__graft_entry__.py", "tests/tools/soak.py", "tests/tools/bench_configs.py", "tools/dense_bench.py"):
        ast.parse((ROOT / f).read_text())
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--direct", "--no-sharded-leg"):
        assert flag in out.stdout
