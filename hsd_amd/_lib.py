"""ctypes loader for libhsd_amd.so (the C ABI declared in include/hsd.h).

There is NO CPU fallback: if the library is missing or cannot be loaded the
import of anything that computes raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HSD_LIB") or os.path.join(_HERE, "libhsd_amd.so")     # HSD_LIB: A/B builds of the library

# status codes (include/hsd.h)
OK = 0
ERR_INVALID = -1
ERR_HIP = -2
ERR_NO_GRAPH = -3
ERR_NO_SIGNATURE = -4
ERR_EMPTY_RING = -5
ERR_TOO_LARGE = -6
ERR_COMM = -7
ERR_NOT_NORMALISED = -8
ERR_UNSUPPORTED = -9
ERR_NOMEM = -10
ERR_INTERNAL = -11

METRIC_WASSERSTEIN = 0
METRIC_KL = 1
METRIC_SYMMETRIC_KL = 2
ALIGN_NONE = 0
ALIGN_ZEROPAD = 1
ABI_VERSION = 2          # HSD_ABI_VERSION of include/hsd.h this binding was written against


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("lds_bytes_per_cu", C.c_int32), ("hbm_bytes", C.c_int64), ("clock_khz", C.c_int32)]


_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int

# every symbol include/hsd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "hsd_abi_version": (_int, []),
    "hsd_create": (_int, [_int, C.POINTER(_p)]),
    "hsd_destroy": (None, [_p]),
    "hsd_last_error": (C.c_char_p, [_p]),
    "hsd_strerror": (C.c_char_p, [_int]),
    "hsd_get_device_info": (_int, [_p, C.POINTER(DeviceInfo)]),
    "hsd_sync": (_int, [_p]),
    "hsd_malloc": (_int, [_p, C.c_size_t, C.POINTER(_p)]),
    "hsd_free": (_int, [_p, _p]),
    "hsd_memcpy_d2h": (_int, [_p, _p, _p, C.c_size_t]),
    "hsd_memcpy_h2d": (_int, [_p, _p, _p, C.c_size_t]),
    "hsd_memset": (_int, [_p, _p, _int, C.c_size_t]),
    "hsd_timer_start": (_int, [_p]),
    "hsd_timer_stop": (_int, [_p, C.POINTER(C.c_float)]),
    "hsd_set_graph": (_int, [_p, _i64, _p, _p]),
    "hsd_khop_sizes": (_int, [_p, _int, _i64, _i64, _p]),
    "hsd_khop_members": (_int, [_p, _int, _i64, _i64, _p, _p]),
    "hsd_signature_build": (_int, [_p, _int, _p, _int, _p, _i64, _i64]),
    "hsd_signature_build_degree": (_int, [_p, _int, _i64, _i64]),
    "hsd_signature_dims": (_int, [_p, C.POINTER(_int), C.POINTER(_int)]),
    "hsd_signature_support": (_int, [_p, _p]),
    "hsd_signature_get_counts": (_int, [_p, _i64, _i64, _p]),
    "hsd_signature_set_counts": (_int, [_p, _i64, _i64, _p]),
    "hsd_comm_unique_id": (_int, [_p]),
    "hsd_comm_init": (_int, [_p, _int, _int, _p]),
    "hsd_comm_shard": (_int, [_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "hsd_signature_allgather": (_int, [_p]),
    "hsd_pair_prepare": (_int, [_p, _int, _int, _int, _int, C.c_double, _int]),
    "hsd_pair_rows": (_int, [_p, _i64, _i64, _p, _i64, _int]),
    "hsd_pair_topk": (_int, [_p, _int, _i64, _i64, _p, _p]),
    "hsd_pair_shard": (_int, [_p, _int, _int, C.POINTER(_i64), C.POINTER(_i64)]),
    "hsd_pair_operand_dims": (_int, [_p, C.POINTER(_int), C.POINTER(_int)]),
    "hsd_pair_work": (_int, [_p, C.POINTER(C.c_double)]),
    "hsd_set_laplacian": (_int, [_p, _i64, _p, _p, _p]),
    "hsd_heat_wavelets": (_int, [_p, _p, _int, C.c_double, C.c_double, _i64, _i64, _p, _i64, _int]),
    "hsd_heat_wavelets_multi": (_int, [_p, _p, _int, _int, C.c_double, C.c_double, _p, _i64]),
    "hsd_ring_stats": (_int, [_p, _int, _p, _i64, _i64, _i64, _p, _p]),
    "hsd_multiscale_ring_stats": (_int, [_p, _int, _p, _int, _int, C.c_double, C.c_double, _i64, _i64, _p, _p]),
    "hsd_heat_signature_build": (_int, [_p, _int, _p, _i64]),
    "hsd_pair_heat_rows": (_int, [_p, _int, _int, _i64, _i64, _p, _i64, _int]),
    "hsd_pair_heat_rowform": (_int, [_p, _p, _i64, _i64, _i64, _p, _i64, _int]),
    "hsd_sample_distance": (_int, [_p, _p, _i64, _p, _i64, _int, _p]),
    "hsd_matrix_stats": (_int, [_p, _p, _i64, _i64, _i64, _i64, _p]),
    "hsd_matrix_add": (_int, [_p, _p, _i64, _p, _i64, _i64, _i64]),
}

_lib = None


def load():
    """Load the shared library once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C hsd_amd/csrc`). "
            "hsd_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    rebuild = "rebuild it: `make -C hsd_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`)"
    try:
        lib.hsd_abi_version.restype = _int
        lib.hsd_abi_version.argtypes = []
        version = lib.hsd_abi_version()
    except AttributeError:
        raise ImportError(f"{LIB_PATH} does not export hsd_abi_version: not a libhsd_amd.so; {rebuild}") from None
    if version != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {version}, this package binds version {ABI_VERSION} (include/hsd.h): a stale library; {rebuild}")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError(f"{LIB_PATH} does not export {name} although its ABI version matches: a stale or partial build; {rebuild}") from None
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
