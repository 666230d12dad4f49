"""f64 kernels on a batch, resident (one chunk = the whole batch; StagedBatch) next to the pipelined host call: is the f64 batch
bound by instructions or by balance?  usage: python tools/bench_f64_staged.py [n=20000] [scale=0.5 del=11.5 ext=2.25]   (0.5 / 11.5 / 2.25 is a DYADIC scheme: every score a
multiple of 0.25, exact zeros everywhere, a third of the pairs fill twice; 0.37 / 11.3 / 2.1 is what a re-estimated matrix looks like)"""
import os
os.environ.setdefault("ALN_NO_DYADIC", "1")      # this tool is about the f64 kernels: a dyadic scheme stays on them
import sys, time
import numpy as np
sys.path.insert(0, ".")
from aligner_amd import _ffi, workloads
from aligner_amd.batch import StagedBatch, align_batch
from aligner_amd.matrices import get_blosum62
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
b = workloads.c5_batch(n)
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
de = float(sys.argv[3]) if len(sys.argv) > 3 else 11.5
ex = float(sys.argv[4]) if len(sys.argv) > 4 else 2.25
S = get_blosum62() * scale
sb = StagedBatch(b, _ffi.CORE_LOCAL, de, ex, S, device=0, outputs=_ffi.OUT_SCORE | _ffi.OUT_TRACEBACK)
sb.run(); sb.sync()
sb.enable_timing(True)
t0 = time.perf_counter()
for _ in range(3):
    sb.run()
sb.sync()
dt = (time.perf_counter() - t0) / 3
tm = sb.timing()
res = sb.fetch(want_traceback=False).results
p = res["passes"] & 0xff
print("resident: %d pairs %.3g cells: step %.2f ms = %.1f GCUPS (fill %.2f ms, traceback %.2f ms); passes: %s" % (
    n, b.cells, dt * 1e3, b.cells / dt / 1e9, tm["fill_ms"], tm["traceback_ms"], dict(zip(*np.unique(p, return_counts=True)))))
sb.close()
r = None
for i in range(4):
    t0 = time.perf_counter(); r = align_batch(b, _ffi.CORE_LOCAL, de, ex, S, want_traceback=True, out=r); dt = time.perf_counter() - t0
print("host to host: %.1f ms = %.1f GCUPS" % (dt * 1e3, b.cells / dt / 1e9))
