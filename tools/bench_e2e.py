"""End-to-end rate of aln_align_batch (host buffers in, host buffers out) on the C5 batch -- what a caller of the C ABI sees.
usage: python tools/bench_e2e.py [pairs] [calls]   (env ALN_CHUNK_CELLS overrides the chunk size)"""
import ctypes as C
import os
import sys
import time

import numpy as np

# eight hardware queues for the pipeline's four streams (the library no longer sets this itself; INTEGRATION.md, "Environment")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import _ffi, runtime, workloads  # noqa: E402
from aligner_amd.batch import RESULT_DTYPE  # noqa: E402
from aligner_amd.matrices import get_blosum62  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
want_tb = os.environ.get("E2E_NO_TB") is None
b = workloads.c5_batch(pairs)
S = get_blosum62()
lib = _ffi.load()
outs = _ffi.OUT_SCORE | (_ffi.OUT_TRACEBACK if want_tb else 0)
p, keep = runtime.make_params(_ffi.CORE_LOCAL, 11, 2, S, outputs=outs)
res = np.zeros(len(b), dtype=RESULT_DTYPE)
tb_off, total = b.tb_layout()
tb = np.zeros(max(total, 1), dtype=np.uint8)
ctx = runtime.context(0)
print("pairs %d cells %.4g seq bytes %.1f MB tb bytes %.1f MB" % (pairs, b.cells, len(b.seqs) / 1e6, total / 1e6), flush=True)
times = []
for i in range(calls + 1):
    t0 = time.perf_counter()
    st = lib.aln_align_batch(ctx, C.byref(p), b.seqs.ctypes.data, b.q_off.ctypes.data, b.q_len.ctypes.data, b.t_off.ctypes.data,
                             b.t_len.ctypes.data, len(b), res.ctypes.data, tb.ctypes.data if want_tb else None,
                             tb_off.ctypes.data if want_tb else None)
    dt = time.perf_counter() - t0
    assert st == 0, (st, _ffi.last_error())
    times.append(dt)
    print("call %d%s: %.2f ms  %.1f GCUPS   ok %d" % (i, " (cold pool)" if i == 0 else "", dt * 1e3, b.cells / dt / 1e9,
                                                 int((res["status"] == 0).sum())), flush=True)
if os.environ.get("E2E_JSON"):
    import json
    warm = sorted(times[1:])
    med = warm[len(warm) // 2]
    print("E2E_JSON " + json.dumps({"ms": round(med * 1e3, 3), "gcups": round(b.cells / med / 1e9, 2), "calls": calls,
                                     "pairs_ok": int((res["status"] == 0).sum())}), flush=True)
