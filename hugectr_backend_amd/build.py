"""In-tree build of the native libraries (hipcc, gfx950 only).

    python -m hugectr_backend_amd.build [--force] [--jobs N]

Outputs (git-ignored, shipped to the GPU box by gpurun):
    hugectr_backend_amd/lib/libhps_amd.so         engine + C ABI (include/hps_amd.h)
    hugectr_backend_amd/lib/libtriton_hps.so      Triton backend shell: exports only TRITONBACKEND_*
    hugectr_backend_amd/lib/libtriton_mock_core.so  TEST INFRASTRUCTURE: stand-in for tritonserver's core
"""
from __future__ import annotations

import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent
# HPS_AMD_LIB_DIR / HPS_AMD_EXTRA_FLAGS: an instrumented variant of the libraries next to the product build
# (tests/test_sanitizers.py builds the shell with -fsanitize=address into a scratch directory this way)
_ALT = os.environ.get("HPS_AMD_LIB_DIR")
LIB = Path(_ALT) if _ALT else PKG / "lib"
OBJ = Path(_ALT) / "obj" if _ALT else PKG / "build"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-pthread",
            f"-I{ROOT / 'include'}", f"-I{CSRC}", *os.environ.get("HPS_AMD_EXTRA_FLAGS", "").split()]

ENGINE_SRCS = [
    "common/json.cpp",
    "common/config.cpp",
    "ps/thread_pool.cpp",
    "ps/host_table.cpp",
    "ps/volatile_tier.cpp",
    "ps/update_source.cpp",
    "cache/kernels.hip",
    "cache/shard_kernels.hip",
    "cache/shard_session.cpp",
    "cache/entry_kernels.hip",
    "cache/shard_entry.cpp",
    "cache/direct_kernels.hip",
    "cache/copy_engines.cpp",
    "cache/segcopy_kernels.hip",
    "cache/probe_kernels.hip",
    "cache/multi_gpu_probe.cpp",
    "cache/engine.cpp",
    "cache/parameter_server.cpp",
    "dense/dense_kernels.hip",
    "dense/fused_kernels.hip",
    "dense/dense.cpp",
]
CAPI_SRCS = ["c_api.cpp"]
TRITON_SRCS = [
    "triton/hps.cpp",
    "triton/backend_state.cpp",
    "triton/model_state.cpp",
    "triton/model_instance_state.cpp",
    "triton/timer.cpp",
    "triton/roctx.cpp",
]
MOCK_SRCS = ["mock_triton/mock_core.cpp"]


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(map(str, cmd)) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    return r


def _deps_stamp(src: Path) -> str:
    """Hash of the source and every header under csrc/ + include/ (coarse but safe)."""
    h = hashlib.sha1()
    h.update(src.read_bytes())
    for d in (CSRC, ROOT / "include"):
        for p in sorted(d.rglob("*.h")):
            h.update(p.read_bytes())
    h.update(" ".join(CXXFLAGS).encode())
    return h.hexdigest()


def _compile(rel: str, force: bool) -> Path:
    src = CSRC / rel
    obj = OBJ / (rel.replace("/", "__") + ".o")
    stamp = obj.with_suffix(".stamp")
    want = _deps_stamp(src)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == want:
        return obj
    obj.parent.mkdir(parents=True, exist_ok=True)
    cmd = [HIPCC, f"--offload-arch={ARCH}", *CXXFLAGS, "-c", str(src), "-o", str(obj)]
    _run(cmd)
    stamp.write_text(want)
    return obj


def _link(out: Path, objs, extra=()):
    out.parent.mkdir(parents=True, exist_ok=True)
    _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(out), *map(str, objs), "-pthread", *extra,
          *os.environ.get("HPS_AMD_EXTRA_LDFLAGS", "").split()])


def build(force: bool = False, jobs: int | None = None, verbose: bool = False) -> dict:
    jobs = jobs or min(8, os.cpu_count() or 1)
    groups = {"engine": ENGINE_SRCS, "capi": CAPI_SRCS}
    if (CSRC / "triton" / "hps.cpp").exists():
        groups["triton"] = TRITON_SRCS
    if (CSRC / "mock_triton" / "mock_core.cpp").exists():
        groups["mock"] = MOCK_SRCS
    all_srcs = sorted({s for g in groups.values() for s in g})
    with ThreadPoolExecutor(jobs) as ex:
        objs = dict(zip(all_srcs, ex.map(lambda s: _compile(s, force), all_srcs)))
    outs = {}

    def newer(out: Path, deps) -> bool:
        return force or not out.exists() or any(o.stat().st_mtime > out.stat().st_mtime for o in deps)

    eng = [objs[s] for s in ENGINE_SRCS]
    out = LIB / "libhps_amd.so"
    deps = eng + [objs[s] for s in CAPI_SRCS]
    if newer(out, deps):
        _link(out, deps)
    outs["hps_amd"] = out
    if "triton" in groups:
        out = LIB / "libtriton_hps.so"
        deps = eng + [objs[s] for s in TRITON_SRCS]
        if newer(out, deps):
            # export only TRITONBACKEND_* like the reference's libtriton_hps.ldscript
            _link(out, deps, [f"-Wl,--version-script={CSRC / 'triton' / 'libtriton_hps.ldscript'}"])
        outs["triton_hps"] = out
    if "mock" in groups:
        out = LIB / "libtriton_mock_core.so"
        deps = [objs[s] for s in MOCK_SRCS]
        if newer(out, deps):
            _link(out, deps, ["-ldl"])
        outs["mock_core"] = out
    # native tools next to the libraries (binaries are git-ignored like the libraries; they travel with gpurun)
    tool_src = ROOT / "tools" / "triton_abi_bench.cpp"
    if tool_src.exists() and "mock" in groups:
        out = LIB / "triton_abi_bench.bin"
        if force or not out.exists() or out.stat().st_mtime < max(tool_src.stat().st_mtime,
                                                                   (CSRC / "mock_triton" / "mock_core.h").stat().st_mtime,
                                                                   (CSRC / "common" / "hps_hash.h").stat().st_mtime):
            _run([HIPCC, f"--offload-arch={ARCH}", "-O2", "-std=c++17", "-Wall", "-Wno-unused-result", str(tool_src), "-o", str(out),
                  "-ldl", "-pthread"])
        outs["triton_abi_bench"] = out
    if verbose:
        for k, v in outs.items():
            print(f"{k}: {v}")
    return outs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs, verbose=True)
