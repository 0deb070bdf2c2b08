"""MI355X-native Hierarchical Parameter Server backend (lookup path of triton-inference-server/hugectr_backend).

The product is native: libhps_amd.so (HIP engine + C ABI) and libtriton_hps.so (Triton backend shell).
This package holds the build driver and thin ctypes bindings used by tests and bench.py.
"""
__version__ = "0.1.0"
