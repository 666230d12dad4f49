"""f64 kernels on a batch, resident (one chunk = the whole batch; StagedBatch) next to the pipelined host call: is the f64 batch
bound by instructions or by balance?  usage: python tools/bench_f64_staged.py [n=20000]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from aligner_amd import _ffi, workloads
from aligner_amd.batch import StagedBatch, align_batch
from aligner_amd.matrices import get_blosum62
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
b = workloads.c5_batch(n)
S = get_blosum62() * 0.5
sb = StagedBatch(b, _ffi.CORE_LOCAL, 11.5, 2.25, S, device=0, outputs=_ffi.OUT_SCORE | _ffi.OUT_TRACEBACK)
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
    t0 = time.perf_counter(); r = align_batch(b, _ffi.CORE_LOCAL, 11.5, 2.25, S, want_traceback=True, out=r); dt = time.perf_counter() - t0
print("host to host: %.1f ms = %.1f GCUPS" % (dt * 1e3, b.cells / dt / 1e9))
