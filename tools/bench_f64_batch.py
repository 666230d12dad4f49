"""f64 kernels on a batch: the first `n` C5 pairs with a real-valued matrix (BLOSUM62 x 0.5, del 11.5 / ext 2.25) through
aln_align_batch, host buffers in and out.  usage: python tools/bench_f64_batch.py [n=4000]"""
import os
os.environ.setdefault("ALN_NO_DYADIC", "1")      # this tool is about the f64 kernels: a dyadic scheme stays on them
import sys, time
sys.path.insert(0, ".")
from aligner_amd import _ffi, workloads
from aligner_amd.batch import align_batch
from aligner_amd.matrices import get_blosum62
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
b = workloads.c5_batch(n)
S = get_blosum62() * 0.5
r = None
for i in range(4):
    t0 = time.perf_counter(); r = align_batch(b, _ffi.CORE_LOCAL, 11.5, 2.25, S, want_traceback=True, out=r); dt = time.perf_counter() - t0
print("f64 batch, %d C5 pairs (%.3g cells): %.1f ms = %.1f GCUPS host to host, %d ok" % (n, b.cells, dt * 1e3, b.cells / dt / 1e9, int((r.results["status"] == 0).sum())))
