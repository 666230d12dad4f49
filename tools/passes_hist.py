"""Histogram of aln_pair_result.passes over a C5 sample: how many pairs took the row-1 repair / a second full pass."""
import sys, numpy as np
sys.path.insert(0,'.')
from aligner_amd import _ffi, workloads
from aligner_amd.batch import StagedBatch
from aligner_amd.matrices import get_blosum62
b = workloads.c5_batch(20000)   # passes: see include/aligner_hip.h (bits 16-19: 1 + checkpoint where the repair stopped)
sb = StagedBatch(b, _ffi.CORE_LOCAL, 11, 2, get_blosum62(), outputs=3)
sb.run(); sb.sync()
r = sb.fetch(False).results
vals, cnt = np.unique(r["passes"], return_counts=True)
print({hex(int(v)): int(c) for v, c in zip(vals, cnt)})
print("flags", np.unique(r["flags"], return_counts=True))
