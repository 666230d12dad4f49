"""Randomized differential run: GPU (through the C ABI) vs the CPU oracle on random shapes, semantics, gap costs and
scoring -- summaries, both aligned strings and every direction.  usage: python tools/fuzz_parity.py [cases [seed]]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi, runtime
from aligner_amd.errors import ReferencePanic
from aligner_amd.matrices import get_blosum62

args = [a for a in sys.argv[1:] if not a.startswith('-')]
cases = int(args[0]) if len(args) > 0 else 300
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 12345)
B62 = get_blosum62()
S4 = np.where(np.eye(24) > 0, 1.0, -1.0)
bad = 0
for c in range(cases):
    sem = int(rng.integers(0, 4))
    big = rng.random() < 0.25                      # some cases large enough for the single-pair route
    N = int(rng.integers(64, 2600)) if big else int(rng.integers(1, 400))
    M = int(rng.integers(130, 1500)) if big else int(rng.integers(1, 700))
    zero_rich = rng.random() < 0.5
    A = 4 if zero_rich else 20
    S = S4 if zero_rich else B62
    dele, ext = [(2, 1), (1, 2), (11, 2), (3, 3), (11, 1), (5, 4)][int(rng.integers(0, 6))]
    if sem >= 2:
        ext = dele
    q = rng.integers(0, A, N).astype(np.uint8)
    t = rng.integers(0, A, M).astype(np.uint8)
    if rng.random() < 0.5 and min(N, M) > 8:       # related sequences: long alignments
        L = min(N, M) // 2
        t[M // 4:M // 4 + L] = q[N // 4:N // 4 + L][:len(t[M // 4:M // 4 + L])]
    if '-v' in sys.argv:
        print('case', c, 'sem', sem, 'N', N, 'M', M, 'gaps', dele, ext, 'zero_rich', zero_rich, flush=True)
    ref = orc.align(sem, q, t, dele, ext, S, want_matrices=True)
    try:
        res, qa, ta, D, H = runtime.align_pair(sem, q, t, dele, ext, S, want_directions=True, want_h=False)
    except ReferencePanic as e:
        ok = ref["status"] == e.status
        if not ok:
            bad += 1; print("MISMATCH status", c, sem, N, M, dele, ext, e.status, ref["status"], flush=True)
        continue
    ok = (ref["status"] == 0 and res.score == ref["score"] and res.f == ref["f"] and (res.end_y, res.end_x) == ref["end"]
          and (res.start_y, res.start_x) == ref["start"] and qa.tolist() == ref["qa"].tolist()
          and ta.tolist() == ref["ta"].tolist() and (D == ref["D"]).all())
    if not ok:
        bad += 1
        print("MISMATCH", c, "sem", sem, "N", N, "M", M, "gaps", dele, ext, "zero_rich", zero_rich, "flags", res.flags,
              "passes", hex(res.passes), flush=True)
    if c % 50 == 49:
        print("case", c + 1, "mismatches", bad, flush=True)
print("done:", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
