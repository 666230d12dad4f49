"""Batches of large pairs through the C ABI, host buffers in and out (aln_align_batch): the regime where the batch kernel shares
the strips of a pair between waves (cooperative passes) instead of sending every pair down the single-pair route one after the other.
usage: python tools/bench_bigpairs.py [calls]   (env ALN_NO_COOP / ALN_BIG_TO_SINGLE=1 select the older behaviour)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import _ffi, runtime, workloads  # noqa: E402
from aligner_amd.batch import RESULT_DTYPE  # noqa: E402
from aligner_amd.matrices import get_blosum62  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = get_blosum62()
lib = _ffi.load()
ctx = runtime.context(0)
for n, L in ((256, 4200), (2000, 4200), (16, 10000), (64, 2000), (1024, 3000)):
    b = workloads.c5_batch(n_pairs=n, lo=L, hi=L)
    p, keep = runtime.make_params(_ffi.CORE_LOCAL, 11, 2, S, outputs=_ffi.OUT_SCORE | _ffi.OUT_TRACEBACK)
    res = np.zeros(len(b), dtype=RESULT_DTYPE)
    tb_off, total = b.tb_layout()
    tb = np.zeros(max(total, 1), dtype=np.uint8)
    best = 1e9
    for i in range(calls + 1):
        t0 = time.perf_counter()
        st = lib.aln_align_batch(ctx, C.byref(p), b.seqs.ctypes.data, b.q_off.ctypes.data, b.q_len.ctypes.data, b.t_off.ctypes.data,
                                 b.t_len.ctypes.data, len(b), res.ctypes.data, tb.ctypes.data, tb_off.ctypes.data)
        dt = time.perf_counter() - t0
        assert st == 0, (st, _ffi.last_error())
        if i:
            best = min(best, dt)
    single = int(((res["flags"] & 2) != 0).sum())
    print("%5d pairs of %5d x %5d: %8.3f ms per call = %7.1f GCUPS (%.2f ms per pair); %d on the single-pair route, %d re-filled" % (
        n, L, L, best * 1e3, b.cells / best / 1e9, best * 1e3 / n, single, int(((res["passes"] & 0xff) >= 2).sum())), flush=True)
