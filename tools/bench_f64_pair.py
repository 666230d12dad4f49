import os
os.environ.setdefault("ALN_NO_DYADIC", "1")      # this tool is about the f64 kernels: a dyadic scheme stays on them
import sys, time
sys.path.insert(0, '.')
import numpy as np
from aligner_amd import _ffi, runtime, workloads
from aligner_amd.matrices import get_blosum62
q, t = workloads.c2_pair(homolog=False)
S = get_blosum62() * 0.5
for N in (1000, 330):
    qq, tt = q[:N], t[:min(N, 300) if N == 330 else N]
    for i in range(3):
        runtime.align_pair(_ffi.CORE_LOCAL, qq, tt, 11.5, 2.25, S)
    ts = []
    for i in range(20):
        t0 = time.perf_counter(); res = runtime.align_pair(_ffi.CORE_LOCAL, qq, tt, 11.5, 2.25, S)[0]; ts.append(time.perf_counter() - t0)
    ts.sort()
    print("f64 pair %d x %d: median %.3f ms min %.3f ms  flags %d passes %#x score %.2f  %.2f GCUPS" % (len(qq), len(tt), ts[10] * 1e3, ts[0] * 1e3, res.flags, res.passes, res.score, len(qq) * len(tt) / ts[10] / 1e9))
