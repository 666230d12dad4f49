"""ctypes binding of the C ABI in include/aligner_hip.h (aligner_amd/lib/libaligner_hip.so).

This is the only door between the Python host mirror and the native HIP path.  It fails loudly when the shared
library is missing: there is no Python or CPU implementation of the DP behind it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALN_LIB") or os.path.join(_HERE, "lib", "libaligner_hip.so")

# enum aln_semantics
CORE_GLOBAL, CORE_LOCAL, LEGACY_GLOBAL, LEGACY_LOCAL, PWM_LOCAL = 0, 1, 2, 3, 4
# enum aln_status
OK = 0
ERR_UNNECESSARY_ARGUMENT = 1
ERR_EMPTY_SEQUENCE = 2
ERR_CODE_OUT_OF_RANGE = 3
ERR_NO_POSITIVE_CELL = 4
ERR_DEVICE = 5
ERR_OOM = 6
ERR_INVALID_ARGUMENT = 7
ERR_UNSUPPORTED = 8
ERR_MATRIX_SHAPE = 9
STATUS_NAMES = {0: "OK", 1: "ERR_UNNECESSARY_ARGUMENT", 2: "ERR_EMPTY_SEQUENCE", 3: "ERR_CODE_OUT_OF_RANGE",
                4: "ERR_NO_POSITIVE_CELL", 5: "ERR_DEVICE", 6: "ERR_OOM", 7: "ERR_INVALID_ARGUMENT",
                8: "ERR_UNSUPPORTED", 9: "ERR_MATRIX_SHAPE"}
# enum aln_outputs
OUT_SCORE, OUT_TRACEBACK, OUT_DIRECTIONS, OUT_H_MATRIX = 1, 2, 4, 8

# every symbol include/aligner_hip.h declares
EXPORTS = [
    "aln_create", "aln_create_multi", "aln_device_count", "aln_destroy", "aln_last_error", "aln_abi_version", "aln_device_info", "aln_align_pair",
    "aln_align_batch", "aln_plan_chunks", "aln_batch_create", "aln_batch_run", "aln_batch_sync", "aln_batch_fetch",
    "aln_batch_destroy", "aln_batch_cells", "aln_batch_size", "aln_batch_results_device",
    "aln_batch_direction_bytes", "aln_batch_timing", "aln_batch_enable_timing",
]


class Params(C.Structure):
    _fields_ = [("semantics", C.c_int32), ("heuristics_present", C.c_int32), ("del_", C.c_double),
                ("ext", C.c_double), ("matrix", C.c_void_p), ("rows", C.c_uint32), ("cols", C.c_uint32),
                ("row_stride", C.c_int64), ("outputs", C.c_uint32), ("blank_code", C.c_uint8),
                ("force_f64", C.c_uint8), ("force_serial", C.c_uint8), ("force_generic", C.c_uint8),
                ("max_passes", C.c_uint32)]


class PairResult(C.Structure):
    _fields_ = [("f", C.c_double), ("score", C.c_double), ("end_y", C.c_uint32), ("end_x", C.c_uint32),
                ("start_y", C.c_uint32), ("start_x", C.c_uint32), ("aln_len", C.c_uint32), ("status", C.c_int32),
                ("passes", C.c_uint32), ("flags", C.c_uint32)]


assert C.sizeof(PairResult) == 48

_lib = None


def load():
    """Loads the native library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "aligner_amd: native library %s is missing -- run `python -m aligner_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the DP path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i, u64p = C.c_void_p, C.c_int, C.c_void_p
    lib.aln_create.restype = vp
    lib.aln_create.argtypes = [i, C.POINTER(C.c_int)]
    lib.aln_create_multi.restype = vp
    lib.aln_create_multi.argtypes = [i, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.aln_device_count.restype = i
    lib.aln_device_count.argtypes = [vp]
    lib.aln_destroy.restype = None
    lib.aln_destroy.argtypes = [vp]
    lib.aln_last_error.restype = C.c_char_p
    lib.aln_last_error.argtypes = []
    lib.aln_abi_version.restype = i
    lib.aln_device_info.restype = i
    lib.aln_device_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    lib.aln_align_pair.restype = i
    lib.aln_align_pair.argtypes = [vp, C.POINTER(Params), vp, C.c_size_t, vp, C.c_size_t, C.POINTER(PairResult), vp,
                                   vp, vp, vp]
    lib.aln_align_batch.restype = i
    lib.aln_align_batch.argtypes = [vp, C.POINTER(Params), vp, u64p, u64p, u64p, u64p, C.c_size_t, vp, vp, u64p]
    lib.aln_plan_chunks.restype = C.c_size_t
    lib.aln_plan_chunks.argtypes = [C.POINTER(Params), u64p, u64p, C.c_size_t, i, u64p, u64p, C.c_size_t]
    lib.aln_batch_create.restype = vp
    lib.aln_batch_create.argtypes = [vp, C.POINTER(Params), vp, u64p, u64p, u64p, u64p, C.c_size_t,
                                     C.POINTER(C.c_int)]
    lib.aln_batch_run.restype = i
    lib.aln_batch_run.argtypes = [vp, vp]
    lib.aln_batch_sync.restype = i
    lib.aln_batch_sync.argtypes = [vp]
    lib.aln_batch_fetch.restype = i
    lib.aln_batch_fetch.argtypes = [vp, vp, vp, u64p]
    lib.aln_batch_destroy.restype = None
    lib.aln_batch_destroy.argtypes = [vp]
    lib.aln_batch_cells.restype = C.c_uint64
    lib.aln_batch_cells.argtypes = [vp]
    lib.aln_batch_size.restype = C.c_size_t
    lib.aln_batch_size.argtypes = [vp]
    lib.aln_batch_results_device.restype = vp
    lib.aln_batch_results_device.argtypes = [vp]
    lib.aln_batch_direction_bytes.restype = C.c_uint64
    lib.aln_batch_direction_bytes.argtypes = [vp]
    lib.aln_batch_timing.restype = i
    lib.aln_batch_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    lib.aln_batch_enable_timing.restype = None
    lib.aln_batch_enable_timing.argtypes = [vp, i]
    _lib = lib
    return lib


def last_error():
    return (load().aln_last_error() or b"").decode("utf-8", "replace")
