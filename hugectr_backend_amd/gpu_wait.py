"""Wait until the HIP runtime reports a device.

Right after another process let go of the GPU, a freshly started process can be told "no device" for a moment (seen on
the MI355X box between back-to-back runs).  A runtime that initialised that way may keep saying so for the life of the
process, so the question is asked from throw-away processes first; callers then initialise HIP/torch themselves.
Nothing here falls back to a CPU path: if no device shows up the caller fails as before.
"""
from __future__ import annotations

import subprocess
import sys
import time

_PROBE = ("import ctypes; l = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so'); n = ctypes.c_int(0); "
          "r = l.hipGetDeviceCount(ctypes.byref(n)); print(n.value if r == 0 else 0)")


def wait_for_gpu(timeout_s: float = 30.0) -> int:
    """Number of HIP devices seen by a fresh process, polled for up to `timeout_s` seconds (0 if none appeared)."""
    deadline = time.time() + timeout_s
    while True:
        try:
            out = subprocess.run([sys.executable, "-c", _PROBE], capture_output=True, text=True, timeout=60).stdout.strip()
            if out.isdigit() and int(out) > 0:
                return int(out)
        except Exception:  # noqa: BLE001
            pass
        if time.time() >= deadline:
            return 0
        time.sleep(1.0)


_BUS_PROBE = ("import ctypes; l = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so'); n = ctypes.c_int(0); "
              "r = l.hipGetDeviceCount(ctypes.byref(n)); b = ctypes.create_string_buffer(64); "
              "print(' '.join((b.value.decode().lower() if l.hipDeviceGetPCIBusId(b, 64, d) == 0 else '?') for d in range(n.value if r == 0 else 0)))")


def gpu_numa_nodes() -> list[int]:
    """NUMA node of every HIP device (-1 where the kernel does not say), asked from a throw-away process so that the caller can
    still choose its CPU affinity BEFORE it starts the HIP runtime (whose threads inherit the mask)."""
    try:
        out = subprocess.run([sys.executable, "-c", _BUS_PROBE], capture_output=True, text=True, timeout=60).stdout.split()
    except Exception:  # noqa: BLE001
        return []
    nodes = []
    for bus in out:
        try:
            nodes.append(int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read()))
        except (OSError, ValueError):
            nodes.append(-1)
    return nodes


def bind_process_to_numa_node(node: int) -> bool:
    """The calling thread — and every thread it creates from here on — runs on the CPUs of `node` (what `numactl --cpunodebind`
    does for a whole process when called before anything else started threads).  Memory follows by first touch."""
    import os
    try:
        spec = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
        cpus = set()
        for part in spec.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) < 2:
            return False
        os.sched_setaffinity(0, cpus)
        return True
    except (OSError, ValueError):
        return False
