import sys, time
import numpy as np
sys.path.insert(0, ".")
from aligner_amd import _ffi, runtime
from aligner_amd.matrices import get_blosum62
S = get_blosum62()
rng = np.random.default_rng(1)
for N, M in ((110000, 700), (131200, 130), (200000, 5000), (120000, 20000)):
    q = rng.integers(0, 20, N).astype(np.uint8); t = rng.integers(0, 20, M).astype(np.uint8)
    runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, S)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); res = runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, S)[0]; ts.append(time.perf_counter() - t0)
    print("%d x %d: %.2f ms = %.1f GCUPS, flags %d passes %#x" % (N, M, min(ts) * 1e3, N * M / min(ts) / 1e9, res.flags, res.passes), flush=True)
