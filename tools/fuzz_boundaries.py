"""Differential sweep over the sizes where the kernels change regime: lane / strip / quad / chunk boundaries in both
dimensions (1, 63, 64, 65, 127 ... 2049 columns; 1 ... 1025 rows), every semantics, vs the CPU oracle (full D)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi, runtime
from aligner_amd.errors import ReferencePanic
from aligner_amd.matrices import get_blosum62

B62 = get_blosum62()
rng = np.random.default_rng(2024)
NS = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 1985, 1986, 2047, 2048, 2049, 2111, 4033, 4097]
MS = [1, 2, 7, 8, 9, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 575, 1023, 1024, 1025]
bad = n = 0
for sem in range(4):
    for N in NS:
        for M in MS:
            if rng.random() < 0.55:
                continue
            q = rng.integers(0, 20, N).astype(np.uint8)
            t = rng.integers(0, 20, M).astype(np.uint8)
            if min(N, M) > 4:
                L = min(N, M) // 2
                t[:L] = q[:L]
            dele, ext = ((11, 2) if rng.random() < 0.7 else (3, 3)) if sem < 2 else (8, 8)
            n += 1
            print("sem", sem, "N", N, "M", M, flush=True)
            ref = orc.align(sem, q, t, dele, ext, B62, want_matrices=True)
            try:
                res, qa, ta, D, H = runtime.align_pair(sem, q, t, dele, ext, B62, want_directions=True, want_h=False)
            except ReferencePanic as e:
                if e.status != ref["status"]:
                    bad += 1; print("MISMATCH status", e.status, ref["status"], flush=True)
                continue
            ok = (ref["status"] == 0 and res.score == ref["score"] and (res.end_y, res.end_x) == ref["end"]
                  and (res.start_y, res.start_x) == ref["start"] and qa.tolist() == ref["qa"].tolist()
                  and ta.tolist() == ref["ta"].tolist() and (D == ref["D"]).all())
            if not ok:
                bad += 1; print("MISMATCH sem", sem, "N", N, "M", M, "flags", res.flags, flush=True)
print("done:", n, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
