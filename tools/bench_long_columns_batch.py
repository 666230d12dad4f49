import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from aligner_amd import _ffi, runtime
from aligner_amd.batch import PairBatch, align_batch
from aligner_amd.matrices import get_blosum62
S = get_blosum62()
rng = np.random.default_rng(1)
for N, M, reps in ((90000, 5000, 1), (90000, 5000, 4), (60000, 5000, 1)):
    pairs = []
    for _ in range(reps):
        pairs.append((rng.integers(0, 20, N).astype(np.uint8), rng.integers(0, 20, M).astype(np.uint8)))
    b = PairBatch.from_pairs(pairs)
    align_batch(b, _ffi.CORE_LOCAL, 11, 2, S)
    t0 = time.perf_counter(); r = align_batch(b, _ffi.CORE_LOCAL, 11, 2, S); dt = time.perf_counter() - t0
    print("%d x (%d x %d) batch: %.2f ms = %.1f GCUPS flags %s passes %s" % (reps, N, M, dt * 1e3, reps * N * M / dt / 1e9, r.results["flags"].tolist(), [hex(int(x)) for x in r.results["passes"]]), flush=True)
