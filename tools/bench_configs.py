"""The smaller BASELINE.md configurations as timings (they are parity cases in tests/, not bench.py lines):
C2 one 1k x 1k protein pair (core local), C3 10 000 nucleotide read pairs 150 x 150 (core global)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from aligner_amd import _ffi, workloads
from aligner_amd.batch import PairBatch, StagedBatch
from aligner_amd.matrices import get_blosum62, nucleotide_matrix


def timed(sb, cells, reps=20):
    sb.run(); sb.sync()
    sb.enable_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        sb.run()
    sb.sync(); dt = (time.perf_counter() - t0) / reps
    tm = sb.timing()
    return dt * 1e3, tm["fill_ms"], tm["traceback_ms"], cells / dt / 1e9


q, t = workloads.c2_pair(homolog=True)
one = PairBatch.from_pairs([(q, t)])
print("C2 1k x 1k: ms %.3f fill %.3f tb %.3f GCUPS %.2f" % timed(StagedBatch(one, _ffi.CORE_LOCAL, 11, 2, get_blosum62(), outputs=3), one.cells))
b = workloads.c3_batch(10000)
print("C3 10k x 150x150 nt: ms %.3f fill %.3f tb %.3f GCUPS %.1f" % timed(StagedBatch(b, _ffi.CORE_GLOBAL, 10, 1, nucleotide_matrix(), outputs=3), b.cells))
