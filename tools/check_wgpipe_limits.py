"""The one-workgroup route of the generic kernels at its size limits (8192 columns, 2048 rows, the R boundaries) against the oracle."""
import sys
sys.path.insert(0, '.')
import numpy as np
import oracle as orc
from aligner_amd import _ffi, runtime
from aligner_amd.matrices import get_blosum62
rng = np.random.default_rng(77)
S = np.round(get_blosum62() * 0.5 + rng.normal(0, 0.05, (24, 24)), 3)
bad = 0
for (N, M) in ((8192, 2048), (8192, 65), (4097, 1025), (16, 2048), (2049, 513)):
    q = rng.integers(0, 20, N).astype(np.uint8); t = rng.integers(0, 20, M).astype(np.uint8)
    L = min(N, M) // 2; t[:L] = q[:L]
    for sem in (_ffi.CORE_LOCAL, _ffi.CORE_GLOBAL):
        ref = orc.align(sem, q, t, 11.5, 2.25, S, want_matrices=True)
        res, qa, ta, D, H = runtime.align_pair(sem, q, t, 11.5, 2.25, S, want_directions=True)
        ok = (res.status == ref["status"] == 0 and res.score == ref["score"] and (res.end_y, res.end_x) == ref["end"] and (res.start_y, res.start_x) == ref["start"]
              and qa.tolist() == ref["qa"].tolist() and ta.tolist() == ref["ta"].tolist() and (D == ref["D"]).all())
        print(N, M, sem, "ok" if ok else "MISMATCH", "flags", res.flags, "passes", hex(res.passes), flush=True)
        bad += 0 if ok else 1
print("mismatches", bad)
