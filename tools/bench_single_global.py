"""One large pair through the single-pair route with the core-global semantics (Needleman-Wunsch-like, BLOSUM62, 11/2)."""
import sys
sys.path.insert(0, '.')
import torch
from aligner_amd import _ffi, workloads
from aligner_amd.batch import PairBatch, StagedBatch
from aligner_amd.matrices import get_blosum62
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n
q, _ = workloads.c4_pair(False, n=n)
_, t = workloads.c4_pair(False, n=m)
one = PairBatch.from_pairs([(q, t)])
sp = StagedBatch(one, _ffi.CORE_GLOBAL, 11, 2, get_blosum62(), outputs=3)
sp.run(); sp.sync(); sp.enable_timing(True)
for _ in range(5):
    sp.run()
sp.sync()
tm = sp.timing(); r = sp.fetch(False).results[0]
print("global N", n, "M", m, "fill_ms", round(tm["fill_ms"], 3), "tb_ms", round(tm["traceback_ms"], 3),
      "fill GCUPS", round(one.cells / tm["fill_ms"] / 1e6, 1), "score", r["score"], "status", r["status"])
