"""The host side of the engine (configuration parser, host tables, thread pools) under AddressSanitizer +
UndefinedBehaviorSanitizer and under ThreadSanitizer.  The reference has no sanitizer or race-detection job
(SURVEY.md §4/§5); this build adds one.  The driver (tests/sanitize/driver.cpp) is compiled with g++ from the product
sources — no GPU involved; libamdhip64 is linked only because the pinned-table allocator references hipHostMalloc.
"""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "hugectr_backend_amd" / "csrc"
UNITS = ["common/json.cpp", "common/config.cpp", "ps/thread_pool.cpp", "ps/host_table.cpp", "ps/volatile_tier.cpp", "ps/update_source.cpp"]


def _build_cmd(out: Path, san: str):
    return ["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
            f"-I{SRC}", f"-I{ROOT / 'include'}", "-I/opt/rocm/include", str(ROOT / "tests" / "sanitize" / "driver.cpp"),
            *[str(SRC / u) for u in UNITS], "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", str(out)]


@pytest.fixture(scope="module")
def binaries(tmp_path_factory):
    if shutil.which("g++") is None or not Path("/opt/rocm/lib/libamdhip64.so").exists():
        pytest.skip("g++ or libamdhip64 not available")
    d = tmp_path_factory.mktemp("san")
    jobs = {name: subprocess.Popen(_build_cmd(d / name, san), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            for name, san in (("asan", "address,undefined"), ("tsan", "thread"))}
    out = {}
    for name, p in jobs.items():
        log, _ = p.communicate(timeout=600)
        if p.returncode != 0:
            if "cannot find" in log and "san" in log:
                pytest.skip(f"sanitizer runtime missing: {log[-300:]}")
            raise AssertionError(f"sanitizer build {name} failed:\n{log[-2000:]}")
        out[name] = d / name
    return out


def _run(binary, what, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([str(binary), what], capture_output=True, text=True, timeout=600, env=env)
    report = (r.stdout + r.stderr)
    assert r.returncode == 0 and f"{what}: ok" in r.stdout, report[-3000:]
    assert "runtime error:" not in report and "ERROR: AddressSanitizer" not in report and "WARNING: ThreadSanitizer" not in report, \
        report[-3000:]


@pytest.mark.parametrize("what", ["parse", "table", "threads", "tiered", "keypack", "updates"])
def test_host_side_under_asan_ubsan(binaries, what):
    # leak checking is off: the HIP runtime library keeps process-lifetime allocations of its own
    _run(binaries["asan"], what, {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})


@pytest.mark.parametrize("what", ["threads", "tiered", "updates"])
def test_host_side_under_tsan(binaries, what):
    _run(binaries["tsan"], what, {"TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0"})


def test_triton_shell_and_engine_host_code_under_asan(tmp_path):
    """The three product libraries rebuilt with -fsanitize=address (host code; device code untouched) into a scratch
    directory, and the CPU-runnable suites of the Triton shell, the host tier and the configuration parser run against
    them with the ASan runtime preloaded: every request/response/error path the mock core drives, under ASan."""
    import sys
    clang = Path("/opt/rocm/lib/llvm/bin/clang")
    if not clang.exists():
        pytest.skip("ROCm clang not available")
    rt = subprocess.run([str(clang), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not rt or not Path(rt).exists():
        pytest.skip("clang ASan runtime not available")
    env = dict(os.environ, HPS_AMD_LIB_DIR=str(tmp_path),
               HPS_AMD_EXTRA_FLAGS="-fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -g -shared-libsan",
               HPS_AMD_EXTRA_LDFLAGS="-fsanitize=address -shared-libsan")
    b = subprocess.run([sys.executable, "-m", "hugectr_backend_amd.build"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert b.returncode == 0, (b.stdout + b.stderr)[-2000:]
    syms = subprocess.run(["nm", "-D", str(tmp_path / "libtriton_hps.so")], capture_output=True, text=True).stdout
    assert "__asan" in syms, "the scratch build is not instrumented"
    env.update(LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_triton_backend_cpu.py", "tests/test_host_tier.py",
                        "tests/test_config.py", "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    report = r.stdout + r.stderr
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in report, report[-3000:]
