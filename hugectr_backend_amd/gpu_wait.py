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
