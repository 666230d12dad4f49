"""ctypes front-end of the CPU oracle (oracle/aligner_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (aligner_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

TOP, LEFT, DIAGONAL, BEGINNING = 0, 1, 2, 3
CORE_GLOBAL, CORE_LOCAL, LEGACY_GLOBAL, LEGACY_LOCAL, PWM_LOCAL = 0, 1, 2, 3, 4
ERR_MATRIX_SHAPE = 9
OK, ERR_UNNECESSARY_ARGUMENT, ERR_EMPTY_SEQUENCE, ERR_CODE_OUT_OF_RANGE, ERR_NO_POSITIVE_CELL = 0, 1, 2, 3, 4


class Params(C.Structure):
    _fields_ = [("semantics", C.c_int32), ("heuristics_present", C.c_int32), ("del_", C.c_double),
                ("ext", C.c_double), ("matrix", C.c_void_p), ("rows", C.c_uint32), ("cols", C.c_uint32),
                ("row_stride", C.c_int64), ("blank_code", C.c_uint8)]


class Result(C.Structure):
    _fields_ = [("f", C.c_double), ("score", C.c_double), ("end_y", C.c_uint32), ("end_x", C.c_uint32),
                ("start_y", C.c_uint32), ("start_x", C.c_uint32), ("coords", C.c_uint64 * 4),
                ("aln_len", C.c_uint32), ("status", C.c_int32)]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("aligner_oracle.c", "aligner_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_align.restype = C.c_int
        _lib.orc_align_batch.restype = C.c_int
    return _lib


def _params(semantics, del_, ext, matrix, heuristics_present=False, blank=98):
    m = np.ascontiguousarray(matrix, dtype=np.float64)
    p = Params(semantics, int(heuristics_present), float(del_), float(ext), m.ctypes.data, m.shape[0], m.shape[1],
               m.shape[1], blank)
    return p, m


def align(semantics, q, t, del_, ext, matrix, want_matrices=False, heuristics_present=False, blank=98):
    """Returns dict(status, f, score, coords, end, start, qa, ta[, H, D])."""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    N, M = len(q), len(t)
    p, keep = _params(semantics, del_, ext, matrix, heuristics_present, blank)
    res = Result()
    qa = np.zeros(M + N + 2, dtype=np.uint8)
    ta = np.zeros(M + N + 2, dtype=np.uint8)
    H = np.zeros((M + 1, N + 1), dtype=np.float64) if want_matrices else None
    D = np.zeros((M + 1, N + 1), dtype=np.uint8) if want_matrices else None
    lib().orc_align(C.byref(p), q.ctypes.data_as(C.c_void_p), C.c_size_t(N), t.ctypes.data_as(C.c_void_p),
                    C.c_size_t(M), C.byref(res), qa.ctypes.data_as(C.c_void_p), ta.ctypes.data_as(C.c_void_p),
                    H.ctypes.data_as(C.c_void_p) if want_matrices else None,
                    D.ctypes.data_as(C.c_void_p) if want_matrices else None)
    out = dict(status=res.status, f=res.f, score=res.score,
               coords=((res.coords[0], res.coords[1]), (res.coords[2], res.coords[3])),
               end=(res.end_y, res.end_x), start=(res.start_y, res.start_x),
               qa=qa[:res.aln_len].copy(), ta=ta[:res.aln_len].copy())
    if want_matrices:
        out["H"], out["D"] = H, D
    return out


def align_pwm(seq, del_, ext, pwm, want_matrices=False, heuristics_present=False, blank=98):
    """PWMAligner::perform_alignment.  Returns dict(status, f, coords, end, start, numbered, qal[, H, D])."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    Q = len(seq)
    p, keep = _params(PWM_LOCAL, del_, ext, pwm, heuristics_present, blank)
    Wd = keep.shape[1]
    res = Result()
    numbered = np.zeros(Q + Wd + 2, dtype=np.uint32)
    qal = np.zeros(Q + Wd + 2, dtype=np.uint8)
    H = np.zeros((Q + 1, Wd + 1), dtype=np.float64) if want_matrices else None
    D = np.zeros((Q + 1, Wd + 1), dtype=np.uint8) if want_matrices else None
    lib().orc_align_pwm(C.byref(p), seq.ctypes.data_as(C.c_void_p), C.c_size_t(Q), C.byref(res),
                        numbered.ctypes.data_as(C.c_void_p), qal.ctypes.data_as(C.c_void_p),
                        H.ctypes.data_as(C.c_void_p) if want_matrices else None,
                        D.ctypes.data_as(C.c_void_p) if want_matrices else None)
    out = dict(status=res.status, f=res.f, score=res.score,
               coords=((res.coords[0], res.coords[1]), (res.coords[2], res.coords[3])),
               end=(res.end_y, res.end_x), start=(res.start_y, res.start_x),
               numbered=numbered[:res.aln_len].copy(), qal=qal[:res.aln_len].copy())
    if want_matrices:
        out["H"], out["D"] = H, D
    return out


def align_batch(semantics, seqs, q_off, q_len, t_off, t_len, del_, ext, matrix, n_threads=1, want_traceback=True,
                blank=98):
    """Batch driver over a packed code buffer. Returns (results ndarray of Result, tb bytes, tb_off)."""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
    q_len = np.ascontiguousarray(q_len, dtype=np.uint64)
    t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
    t_len = np.ascontiguousarray(t_len, dtype=np.uint64)
    n = len(q_off)
    p, keep = _params(semantics, del_, ext, matrix, False, blank)
    res = (Result * n)()
    tb = tb_off = None
    if want_traceback:
        cap = 2 * (q_len + t_len + 2)
        tb_off = np.zeros(n, dtype=np.uint64)
        tb_off[1:] = np.cumsum(cap)[:-1]
        tb = np.zeros(int(cap.sum()), dtype=np.uint8)
    lib().orc_align_batch(C.byref(p), seqs.ctypes.data_as(C.c_void_p), q_off.ctypes.data_as(C.c_void_p),
                          q_len.ctypes.data_as(C.c_void_p), t_off.ctypes.data_as(C.c_void_p),
                          t_len.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_int(n_threads), res,
                          tb.ctypes.data_as(C.c_void_p) if want_traceback else None,
                          tb_off.ctypes.data_as(C.c_void_p) if want_traceback else None)
    return res, tb, tb_off


def midline(qa, ta, matrix, blank=98, pos=99):
    qa = np.ascontiguousarray(qa, dtype=np.uint8)
    ta = np.ascontiguousarray(ta, dtype=np.uint8)
    m = np.ascontiguousarray(matrix, dtype=np.float64)
    out = np.zeros(len(qa), dtype=np.uint8)
    lib().orc_midline(qa.ctypes.data_as(C.c_void_p), ta.ctypes.data_as(C.c_void_p), C.c_size_t(len(qa)),
                      m.ctypes.data_as(C.c_void_p), C.c_int64(m.shape[1]), C.c_uint8(blank), C.c_uint8(pos),
                      out.ctypes.data_as(C.c_void_p))
    return out


def frequency_matrix(qa, ta, volume, blank=98):
    qa = np.ascontiguousarray(qa, dtype=np.uint8)
    ta = np.ascontiguousarray(ta, dtype=np.uint8)
    out = np.zeros((volume, volume), dtype=np.float64)
    lib().orc_frequency_matrix(qa.ctypes.data_as(C.c_void_p), ta.ctypes.data_as(C.c_void_p), C.c_size_t(len(qa)),
                               C.c_uint8(blank), C.c_uint32(volume), out.ctypes.data_as(C.c_void_p))
    return out
