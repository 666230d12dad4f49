"""One shape of tools/bench_bigpairs.py, for profiling: python tools/bench_bigpairs_one.py [pairs=256] [len=4200] [calls=6]"""
import ctypes as C
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import _ffi, runtime, workloads  # noqa: E402
from aligner_amd.batch import RESULT_DTYPE  # noqa: E402
from aligner_amd.matrices import get_blosum62  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4200
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 6
S = get_blosum62()
lib = _ffi.load()
ctx = runtime.context(0)
b = workloads.c5_batch(n_pairs=n, lo=L, hi=L)
outs = _ffi.OUT_SCORE | (0 if os.environ.get("NO_TB") else _ffi.OUT_TRACEBACK)
p, keep = runtime.make_params(_ffi.CORE_LOCAL, 11, 2, S, outputs=outs)
res = np.zeros(len(b), dtype=RESULT_DTYPE)
tb_off, total = b.tb_layout()
tb = np.zeros(max(total, 1), dtype=np.uint8)
ts = []
for i in range(calls + 1):
    t0 = time.perf_counter()
    st = lib.aln_align_batch(ctx, C.byref(p), b.seqs.ctypes.data, b.q_off.ctypes.data, b.q_len.ctypes.data, b.t_off.ctypes.data,
                             b.t_len.ctypes.data, len(b), res.ctypes.data, tb.ctypes.data if outs & _ffi.OUT_TRACEBACK else None,
                             tb_off.ctypes.data if outs & _ffi.OUT_TRACEBACK else None)
    ts.append(time.perf_counter() - t0)
    assert st == 0
print("%d pairs of %d^2: best %.3f ms median %.3f ms = %.1f GCUPS; aln_len mean %.0f" % (n, L, min(ts[1:]) * 1e3, sorted(ts[1:])[len(ts) // 2 - 1] * 1e3,
      b.cells / min(ts[1:]) / 1e9, res["aln_len"].mean()))
