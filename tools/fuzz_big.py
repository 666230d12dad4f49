"""Randomized differential run of the single-pair route at sizes where R = 2 (M > 4096) and N crosses several tracker
chunks / ring laps: every semantics; summaries and both aligned strings (no full D: the matrices are tens of MB).
usage: python tools/fuzz_big.py [cases [seed]]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi, runtime
from aligner_amd.matrices import get_blosum62

args = [a for a in sys.argv[1:] if not a.startswith('-')]
cases = int(args[0]) if len(args) > 0 else 16
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 7)
B62 = get_blosum62()
S4 = np.where(np.eye(24) > 0, 1.0, -1.0)
bad = 0
for c in range(cases):
    sem = int(rng.integers(0, 4)) if c % 2 else _ffi.CORE_LOCAL
    N = int(rng.integers(64, 9000))
    M = int(rng.integers(4097, 9000)) if rng.random() < 0.6 else int(rng.integers(300, 4096))
    zero_rich = rng.random() < 0.4
    A, S = (4, S4) if zero_rich else (20, B62)
    dele, ext = [(2, 1), (11, 2), (3, 3), (11, 1)][int(rng.integers(0, 4))]
    if sem >= 2:
        ext = dele
    q = rng.integers(0, A, N).astype(np.uint8)
    t = rng.integers(0, A, M).astype(np.uint8)
    if rng.random() < 0.6:
        L = min(N, M) // 2
        t[M // 5:M // 5 + L] = q[N // 5:N // 5 + L]
    print("case", c, "sem", sem, "N", N, "M", M, "gaps", dele, ext, "zero_rich", zero_rich, flush=True)
    ref = orc.align(sem, q, t, dele, ext, S)
    res, qa, ta, D, H = runtime.align_pair(sem, q, t, dele, ext, S)
    ok = (ref["status"] == 0 and res.status == 0 and res.score == ref["score"] and res.f == ref["f"]
          and (res.end_y, res.end_x) == ref["end"] and (res.start_y, res.start_x) == ref["start"]
          and qa.tolist() == ref["qa"].tolist() and ta.tolist() == ref["ta"].tolist())
    print("   ", "ok" if ok else "MISMATCH", "flags", res.flags, "passes", hex(res.passes), "score", res.score, ref["score"], flush=True)
    bad += 0 if ok else 1
print("done:", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
