import sys, numpy as np
sys.path.insert(0,'.')
from aligner_amd import _ffi, workloads
from aligner_amd.batch import StagedBatch
from aligner_amd.matrices import get_blosum62
b = workloads.c5_batch(20000)
sb = StagedBatch(b, _ffi.CORE_LOCAL, 11, 2, get_blosum62(), outputs=3)
sb.run(); sb.sync()
r = sb.fetch(False).results
L = r["aln_len"]
print("aln_len: mean %.1f median %d p90 %d p99 %d max %d" % (L.mean(), np.median(L), np.percentile(L,90), np.percentile(L,99), L.max()))
print("score: mean %.1f max %.0f" % (r["score"].mean(), r["score"].max()))
