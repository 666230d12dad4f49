import os
os.environ.setdefault("ALN_NO_DYADIC", "1")      # this tool is about the f64 kernels: a dyadic scheme stays on them
import sys, time
sys.path.insert(0, '.')
import numpy as np
from aligner_amd import _ffi, runtime
from aligner_amd.matrices import get_blosum62
S = get_blosum62() * 0.5
rng = np.random.default_rng(0)
tot = []
for i in range(12):
    q = rng.integers(0, 20, 1000).astype(np.uint8); t = rng.integers(0, 20, 1000).astype(np.uint8)
    for _ in range(2): runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11.5, 2.25, S)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); r = runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11.5, 2.25, S)[0]; ts.append(time.perf_counter() - t0)
    tot.append(sorted(ts)[4]); print("pair %d: %.3f ms passes %#x" % (i, sorted(ts)[4] * 1e3, r.passes))
print("mean %.3f ms" % (np.mean(tot) * 1e3))
