"""ctypes driver of the mock Triton core (hugectr_backend_amd/csrc/mock_triton) — test infrastructure.

Plays tritonserver for libtriton_hps.so: the mock core library is loaded RTLD_GLOBAL so that the backend's
imported TRITONSERVER_*/TRITONBACKEND_* symbols resolve against it, then the backend is dlopen'ed by the
mock exactly like the real server does.
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
import os as _os  # noqa: E402
LIBDIR = Path(_os.environ["HPS_AMD_LIB_DIR"]) if _os.environ.get("HPS_AMD_LIB_DIR") else ROOT / "hugectr_backend_amd" / "lib"
BACKEND_LIB = LIBDIR / "libtriton_hps.so"
MOCK_LIB = LIBDIR / "libtriton_mock_core.so"

TYPE_INT32, TYPE_INT64, TYPE_FP32 = 8, 9, 11
MEM_CPU, MEM_CPU_PINNED, MEM_GPU = 0, 1, 2
KIND_CPU, KIND_GPU = 1, 2
ERR = {"UNKNOWN": 0, "INTERNAL": 1, "NOT_FOUND": 2, "INVALID_ARG": 3, "UNAVAILABLE": 4, "UNSUPPORTED": 5,
       "ALREADY_EXISTS": 6}

EXPORTS = ["TRITONBACKEND_Initialize", "TRITONBACKEND_Finalize", "TRITONBACKEND_ModelInitialize",
           "TRITONBACKEND_ModelFinalize", "TRITONBACKEND_ModelInstanceInitialize",
           "TRITONBACKEND_ModelInstanceFinalize", "TRITONBACKEND_ModelInstanceExecute"]


class InstanceStats(C.Structure):
    _fields_ = [("success_requests", C.c_uint64), ("failed_requests", C.c_uint64), ("batch_reports", C.c_uint64),
                ("last_batch_size", C.c_uint64), ("last_distinct_compute_starts", C.c_uint64)]


class TritonError(RuntimeError):
    def __init__(self, code_plus_1: int, msg: str):
        super().__init__(f"[triton error {code_plus_1 - 1}] {msg}")
        self.code = code_plus_1 - 1
        self.msg = msg


_L = None


def lib():
    global _L
    if _L is None:
        L = C.CDLL(str(MOCK_LIB), mode=C.RTLD_GLOBAL)
        P, cp = C.c_void_p, C.c_char_p
        L.mock_last_error.restype = cp
        L.mock_server_create.argtypes = [cp, cp, cp, C.c_uint32, C.c_uint32, C.POINTER(P)]
        L.mock_server_destroy.argtypes = [P]
        L.mock_model_load.argtypes = [P, cp, C.c_uint64, cp, C.POINTER(P)]
        L.mock_model_unload.argtypes = [P]
        L.mock_instance_create.argtypes = [P, cp, C.c_int, C.c_int32, C.POINTER(P)]
        L.mock_instance_destroy.argtypes = [P]
        L.mock_request_new.restype = P
        L.mock_request_new.argtypes = [cp, C.c_uint64]
        L.mock_request_delete.argtypes = [P]
        L.mock_request_add_input_buffer.argtypes = [P, cp, C.c_int, P, C.c_uint32, P, C.c_uint64, C.c_int, C.c_int64]
        L.mock_request_add_requested_output.argtypes = [P, cp]
        L.mock_request_set_output_buffer.argtypes = [P, P, C.c_uint64, C.c_int, C.c_int64]
        L.mock_instance_execute.argtypes = [P, P, C.c_uint32]
        for f in ("mock_request_response_count", "mock_request_release_count", "mock_request_response_final",
                  "mock_request_error_code", "mock_request_output_count"):
            getattr(L, f).argtypes = [P]
        L.mock_request_error_message.restype = cp
        L.mock_request_error_message.argtypes = [P]
        L.mock_request_output_name.restype = cp
        L.mock_request_output_name.argtypes = [P, C.c_int]
        L.mock_request_output_datatype.argtypes = [P, C.c_int]
        L.mock_request_output_dims.argtypes = [P, C.c_int, P, C.c_int]
        L.mock_request_output_buffer.restype = P
        L.mock_request_output_buffer.argtypes = [P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int),
                                                 C.POINTER(C.c_int64)]
        L.mock_request_response_int_param.argtypes = [P, cp, C.POINTER(C.c_int64)]
        L.mock_instance_get_stats.argtypes = [P, C.POINTER(InstanceStats)]
        L.mock_server_log_counts.argtypes = [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.mock_set_verbose.argtypes = [C.c_int]
        _L = L
    return _L


def _check(rc):
    if rc != 0:
        raise TritonError(rc, (lib().mock_last_error() or b"").decode(errors="replace"))


def model_config(name="hps_model", gpus=(0,), count=1, kind="KIND_GPU", max_batch_size=1024, parameters=None,
                 inputs=None, outputs=None) -> dict:
    """The JSON Triton derives from the sample config.pbtxt
    (/root/reference/hps_backend/samples/Hierarchical_Parameter_Server_Deployment.ipynb:213-245)."""
    cfg = {
        "name": name, "backend": "hps", "max_batch_size": max_batch_size,
        "input": inputs if inputs is not None else [
            {"name": "KEYS", "data_type": "TYPE_INT64", "dims": [-1]},
            {"name": "NUMKEYS", "data_type": "TYPE_INT32", "dims": [-1]}],
        "output": outputs if outputs is not None else [{"name": "OUTPUT0", "data_type": "TYPE_FP32", "dims": [-1]}],
        "instance_group": [{"count": count, "kind": kind, "gpus": list(gpus)}],
    }
    if parameters:
        cfg["parameters"] = {k: {"string_value": str(v)} for k, v in parameters.items()}
    return cfg


class Request:
    def __init__(self, rid="1", correlation_id=0):
        self.L = lib()
        self._h = self.L.mock_request_new(str(rid).encode(), correlation_id)
        self.id = str(rid)
        self._keep = []

    def add_input(self, name, array: np.ndarray, shape=None, dtype_code=None):
        a = np.ascontiguousarray(array)
        self._keep.append(a)
        shp = np.asarray(shape if shape is not None else a.shape, dtype=np.int64)
        self._keep.append(shp)
        code = dtype_code if dtype_code is not None else {np.dtype(np.int64): TYPE_INT64, np.dtype(np.int32): TYPE_INT32,
                                                          np.dtype(np.float32): TYPE_FP32}[a.dtype]
        _check(self.L.mock_request_add_input_buffer(self._h, name.encode(), code, shp.ctypes.data, shp.size,
                                                    a.ctypes.data, a.nbytes, MEM_CPU, 0))
        return self

    def add_input_raw(self, name, dtype_code, shape, ptr, nbytes, memory_type, memory_type_id=0):
        shp = np.asarray(shape, dtype=np.int64)
        self._keep.append(shp)
        _check(self.L.mock_request_add_input_buffer(self._h, name.encode(), dtype_code, shp.ctypes.data, shp.size,
                                                    C.c_void_p(ptr), nbytes, memory_type, memory_type_id))
        return self

    def request_output(self, name="OUTPUT0"):
        _check(self.L.mock_request_add_requested_output(self._h, name.encode()))
        return self

    def set_output_buffer(self, ptr, nbytes, memory_type, memory_type_id=0, keep=None):
        if keep is not None:
            self._keep.append(keep)
        _check(self.L.mock_request_set_output_buffer(self._h, C.c_void_p(ptr), nbytes, memory_type, memory_type_id))
        return self

    # ---- results ----
    @property
    def response_count(self): return self.L.mock_request_response_count(self._h)
    @property
    def release_count(self): return self.L.mock_request_release_count(self._h)
    @property
    def final(self): return bool(self.L.mock_request_response_final(self._h))
    @property
    def error_code(self): return self.L.mock_request_error_code(self._h)
    @property
    def error_message(self): return (self.L.mock_request_error_message(self._h) or b"").decode()
    @property
    def output_count(self): return self.L.mock_request_output_count(self._h)

    def output(self, index=0):
        """(name, datatype, shape, ptr, nbytes, memory_type, memory_type_id)"""
        name = self.L.mock_request_output_name(self._h, index).decode()
        dt = self.L.mock_request_output_datatype(self._h, index)
        shp = (C.c_int64 * 8)()
        nd = self.L.mock_request_output_dims(self._h, index, shp, 8)
        nbytes, mt, mid = C.c_uint64(), C.c_int(), C.c_int64()
        ptr = self.L.mock_request_output_buffer(self._h, index, C.byref(nbytes), C.byref(mt), C.byref(mid))
        return name, dt, list(shp[:nd]), ptr, nbytes.value, mt.value, mid.value

    def output_numpy(self, index=0) -> np.ndarray:
        name, dt, shp, ptr, nbytes, mt, _ = self.output(index)
        assert dt == TYPE_FP32 and mt in (MEM_CPU, MEM_CPU_PINNED)
        n = nbytes // 4
        if n == 0:
            return np.zeros(0, np.float32)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,)).copy()

    def int_param(self, name):
        v = C.c_int64()
        return v.value if self.L.mock_request_response_int_param(self._h, name.encode(), C.byref(v)) == 0 else None

    def close(self):
        if self._h:
            self.L.mock_request_delete(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Instance:
    def __init__(self, model, name, kind, device_id):
        self.L = lib()
        self.model = model
        h = C.c_void_p()
        _check(self.L.mock_instance_create(model._h, name.encode(), kind, device_id, C.byref(h)))
        self._h = h

    def execute(self, requests):
        arr = (C.c_void_p * len(requests))(*[r._h for r in requests])
        _check(self.L.mock_instance_execute(self._h, arr, len(requests)))

    def stats(self) -> InstanceStats:
        s = InstanceStats()
        _check(self.L.mock_instance_get_stats(self._h, C.byref(s)))
        return s

    def destroy(self):
        if self._h:
            h, self._h = self._h, None
            _check(self.L.mock_instance_destroy(h))


class Model:
    def __init__(self, server, name, version, config: dict):
        self.L = lib()
        self.server = server
        h = C.c_void_p()
        _check(self.L.mock_model_load(server._h, name.encode(), version, json.dumps(config).encode(), C.byref(h)))
        self._h = h
        self.instances = []

    def create_instance(self, name=None, kind=KIND_GPU, device_id=0) -> Instance:
        i = Instance(self, name or "inst", kind, device_id)
        self.instances.append(i)
        return i

    def unload(self):
        for i in self.instances:
            i.destroy()
        self.instances = []
        if self._h:
            h, self._h = self._h, None
            _check(self.L.mock_model_unload(h))


class Server:
    """tritonserver --backend-config=hps,ps=<ps_json_path>"""

    def __init__(self, ps_json_path, backend_lib=BACKEND_LIB, api=(0, 0), backend_config=None):
        self.L = lib()
        cfg = backend_config if backend_config is not None else {"cmdline": {"auto-complete-config": "true",
                                                                               "ps": str(ps_json_path)}}
        h = C.c_void_p()
        _check(self.L.mock_server_create(str(backend_lib).encode(), b"hps", json.dumps(cfg).encode(), api[0], api[1],
                                         C.byref(h)))
        self._h = h
        self.models = []

    def load_model(self, name, config: dict, version=1) -> Model:
        m = Model(self, name, version, config)
        self.models.append(m)
        return m

    def log_counts(self):
        i, w, e = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.L.mock_server_log_counts(self._h, C.byref(i), C.byref(w), C.byref(e))
        return i.value, w.value, e.value

    def shutdown(self):
        for m in self.models:
            m.unload()
        self.models = []
        if self._h:
            h, self._h = self._h, None
            _check(self.L.mock_server_destroy(h))
