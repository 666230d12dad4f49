"""Latency of the blocking one-pair entry point (aln_align_pair = one perform_alignment call), host buffers in and out."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from aligner_amd import _ffi, runtime, workloads
from aligner_amd.matrices import get_blosum62
S = get_blosum62()
for n in (100, 1000, 3000):
    q, _ = workloads.c4_pair(False, n=n); _, t = workloads.c4_pair(False, n=n)
    runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, S)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, S)
    dt = (time.perf_counter() - t0) / reps
    print("%d x %d: %.3f ms per call (%.2f GCUPS)" % (n, n, dt * 1e3, n * n / dt / 1e9))
