"""Process-wide context of the native path (one per process per GPU) and the low-level call wrappers.

The reference calls its aligners synchronously on the caller's thread with no shared state
(aligner-core/src/statistics/mod.rs:255-286 runs ten such threads); here every thread shares one `aln_ctx`
bound to this process's GPU.  One process per GPU: LOCAL_RANK picks the device.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _ffi
from .errors import AlignerError, DeviceError, ErrorKind, ReferencePanic

_lock = threading.Lock()
_ctx = {}


def default_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


def context(device=None):
    """Returns the aln_ctx handle for `device`, creating it on first use.  Raises if there is no GPU."""
    lib = _ffi.load()
    dev = default_device() if device is None else int(device)
    with _lock:
        h = _ctx.get(dev)
        if h is None:
            st = C.c_int(0)
            h = lib.aln_create(dev, C.byref(st))
            if not h:
                raise DeviceError(st.value, "aln_create(device=%d) failed: %s [%s] -- the DP path needs an MI355X; "
                                  "there is no CPU fallback" % (dev, _ffi.STATUS_NAMES.get(st.value, st.value),
                                                                _ffi.last_error()))
            _ctx[dev] = h
        return h


def context_multi(devices=None):
    """One context spanning several GPUs of THIS process (aln_create_multi): `devices` is a list of device ids, None = every
    visible device.  Batch calls on it are sharded chunk by chunk over the devices; single calls take the devices in turn.
    (A list may name a device twice: two pools on one GPU -- used by the tests to run the sharded path on a one-GPU box.)"""
    lib = _ffi.load()
    key = ("multi",) + (tuple(int(d) for d in devices) if devices is not None else ("all",))
    with _lock:
        h = _ctx.get(key)
        if h is None:
            st = C.c_int(0)
            if devices is None:
                h = lib.aln_create_multi(0, None, C.byref(st))
            else:
                ids = (C.c_int * len(devices))(*[int(d) for d in devices])
                h = lib.aln_create_multi(len(devices), ids, C.byref(st))
            if not h:
                raise DeviceError(st.value, "aln_create_multi(%s) failed: %s [%s]" % (devices, _ffi.STATUS_NAMES.get(st.value, st.value),
                                                                                  _ffi.last_error()))
            _ctx[key] = h
        return h


def device_info(device=None):
    lib = _ffi.load()
    cus, hbm, name = C.c_int(0), C.c_size_t(0), C.create_string_buffer(128)
    lib.aln_device_info(context(device), C.byref(cus), C.byref(hbm), name, 128)
    return dict(compute_units=cus.value, hbm_bytes=hbm.value, name=name.value.decode())


def raise_for_status(st, where=""):
    """Maps a C-ABI status to the reference's error behaviour."""
    if st == _ffi.OK:
        return
    name = _ffi.STATUS_NAMES.get(st, str(st))
    if st == _ffi.ERR_UNNECESSARY_ARGUMENT:
        raise AlignerError(ErrorKind.UnnecessaryArgument)          # simple/mod.rs:49-51
    if st == _ffi.ERR_MATRIX_SHAPE:
        raise AlignerError(ErrorKind.MatrixShapeError)             # pwm/mod.rs:40-42
    if st == _ffi.ERR_EMPTY_SEQUENCE:
        raise ReferencePanic(st, "called `Option::unwrap()` on a `None` value (empty sequence; simple/mod.rs:103)")
    if st == _ffi.ERR_CODE_OUT_OF_RANGE:
        raise ReferencePanic(st, "ndarray: index out of bounds (residue code outside the matrix; simple/mod.rs:85)")
    if st == _ffi.ERR_NO_POSITIVE_CELL:
        raise ReferencePanic(st, "attempt to subtract with overflow (no positive cell; simple/mod.rs:214)")
    if st in (_ffi.ERR_DEVICE, _ffi.ERR_OOM):
        raise DeviceError(st, "%s%s: %s" % (where and where + ": ", name, _ffi.last_error()))
    raise ValueError("%s%s: %s" % (where and where + ": ", name, _ffi.last_error()))


def make_params(semantics, del_, ext, matrix, heuristics_present=False, outputs=0, blank=98, force_f64=False,
                force_serial=False, force_generic=False, max_passes=0):
    """Returns (Params, keepalive).  `matrix` is the caller's Array2<f64>: any 2-D float array, any row stride."""
    m = np.asarray(matrix, dtype=np.float64)
    if m.ndim != 2:
        raise AlignerError(ErrorKind.MatrixShapeError)
    # (a broadcast / transposed view has a row stride the C side would misread: 0 means "contiguous" there)
    if m.strides[1] != 8 or m.strides[0] % 8 != 0 or m.strides[0] < 8 * m.shape[1]:
        m = np.ascontiguousarray(m)
    p = _ffi.Params(int(semantics), int(bool(heuristics_present)), float(del_), float(ext), m.ctypes.data,
                    m.shape[0], m.shape[1], m.strides[0] // 8, int(outputs), int(blank), int(bool(force_f64)),
                    int(bool(force_serial)), int(bool(force_generic)), int(max_passes))
    return p, m


def align_pair(semantics, query, target, del_, ext, matrix, heuristics_present=False, want_directions=False,
               want_h=False, device=None, **kw):
    """One blocking perform_alignment through aln_align_pair.  Returns (PairResult, qa, ta, D|None, H|None)."""
    lib = _ffi.load()
    q = np.ascontiguousarray(query, dtype=np.uint8)
    t = np.ascontiguousarray(target, dtype=np.uint8)
    N, M = len(q), len(t)
    p, keep = make_params(semantics, del_, ext, matrix, heuristics_present, **kw)
    res = _ffi.PairResult()
    qa = np.zeros(N + M + 2, dtype=np.uint8)
    ta = np.zeros(N + M + 2, dtype=np.uint8)
    D = np.zeros((M + 1, N + 1), dtype=np.uint8) if want_directions else None
    H = np.zeros((M + 1, N + 1), dtype=np.float64) if want_h else None
    st = lib.aln_align_pair(context(device), C.byref(p), q.ctypes.data, N, t.ctypes.data, M, C.byref(res),
                            qa.ctypes.data, ta.ctypes.data, D.ctypes.data if want_directions else None,
                            H.ctypes.data if want_h else None)
    raise_for_status(st, "aln_align_pair")
    return res, qa[:res.aln_len].copy(), ta[:res.aln_len].copy(), D, H


def align_pwm(seq, del_, ext, pwm, heuristics_present=False, want_directions=False, want_h=False, device=None, **kw):
    """One blocking PWMAligner::perform_alignment through aln_align_pair (ALN_PWM_LOCAL).
    Returns (PairResult, numbered u32[], residues u8[], D|None, H|None)."""
    lib = _ffi.load()
    t = np.ascontiguousarray(seq, dtype=np.uint8)
    p, keep = make_params(_ffi.PWM_LOCAL, del_, ext, pwm, heuristics_present, **kw)
    W, M = keep.shape[1], len(t)
    res = _ffi.PairResult()
    numbered = np.zeros(W + M + 2, dtype=np.uint32)
    qal = np.zeros(W + M + 2, dtype=np.uint8)
    D = np.zeros((M + 1, W + 1), dtype=np.uint8) if want_directions else None
    H = np.zeros((M + 1, W + 1), dtype=np.float64) if want_h else None
    st = lib.aln_align_pair(context(device), C.byref(p), None, W, t.ctypes.data, M, C.byref(res),
                            numbered.ctypes.data, qal.ctypes.data, D.ctypes.data if want_directions else None,
                            H.ctypes.data if want_h else None)
    raise_for_status(st, "aln_align_pair")
    return res, numbered[:res.aln_len].copy(), qal[:res.aln_len].copy(), D, H
