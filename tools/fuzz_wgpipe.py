"""Randomized differential run of the one-workgroup-per-pair route of the generic kernels (aln_fill_wgpipe_kernel): single pairs
with a real-valued matrix, forced f64, or integers forced off the fast path; every semantics; shapes across the route's limits
(rows 65..2048: R = 1 / 2 / 4; columns 16..4000: ring wrap-around); zero-rich scoring (several advice passes, strict-order
fall-back) -- summary, both strings and every direction against the CPU oracle, and the route flag (bit 2).
usage: python tools/fuzz_wgpipe.py [cases [seed]]"""
import os
os.environ.setdefault("ALN_NO_DYADIC", "1")      # this tool is about the f64 kernels: a dyadic scheme stays on them
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi, runtime
from aligner_amd.errors import ReferencePanic
from aligner_amd.matrices import get_blosum62

args = [a for a in sys.argv[1:] if not a.startswith('-')]
cases = int(args[0]) if len(args) > 0 else 200
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 2024)
B62 = get_blosum62()
bad = routed = multi = serial = 0
for c in range(cases):
    sem = int(rng.integers(0, 4))
    M = int(rng.choice([int(rng.integers(65, 300)), int(rng.integers(300, 1100)), int(rng.integers(1025, 2049))]))
    N = int(rng.choice([int(rng.integers(16, 300)), int(rng.integers(300, 1500)), int(rng.integers(1500, 4000))]))
    if N * M > 3000000:
        N = max(16, 3000000 // M)
    zero_rich = rng.random() < 0.4
    A = 4 if zero_rich else 20
    mode = ["real", "f64", "generic"][int(rng.integers(0, 3))] if sem < 2 else "generic"
    S = (np.where(np.eye(24) > 0, 1.0, -1.0) if zero_rich else B62).copy()
    dele, ext = [(2, 1), (1, 2), (11, 2), (3, 3), (11, 1), (5, 4)][int(rng.integers(0, 6))]
    if mode == "real":
        S = np.round(S * 0.5 + (0 if zero_rich else rng.normal(0, 0.05, S.shape)), 3)
        dele, ext = dele + 0.5, ext + 0.25
    if sem >= 2:
        ext = dele
    q = rng.integers(0, A, N).astype(np.uint8)
    t = rng.integers(0, A, M).astype(np.uint8)
    if rng.random() < 0.5 and min(N, M) > 8:
        L = min(N, M) // 2
        seg = q[N // 4:N // 4 + L]
        t[M // 4:M // 4 + len(seg)][:] = seg[:len(t[M // 4:M // 4 + len(seg)])]
    kw = dict(force_f64=True) if mode == "f64" else dict(force_generic=True) if mode == "generic" else {}
    if '-v' in sys.argv:
        print('case', c, 'sem', sem, mode, 'N', N, 'M', M, 'gaps', dele, ext, 'zero_rich', zero_rich, flush=True)
    ref = orc.align(sem, q, t, dele, ext, S, want_matrices=True)
    try:
        res, qa, ta, D, H = runtime.align_pair(sem, q, t, dele, ext, S, want_directions=True, want_h=False, **kw)
    except ReferencePanic as e:
        if ref["status"] != e.status:
            bad += 1; print("MISMATCH status", c, sem, N, M, e.status, ref["status"], flush=True)
        continue
    ok = (ref["status"] == 0 and res.score == ref["score"] and res.f == ref["f"] and (res.end_y, res.end_x) == ref["end"]
          and (res.start_y, res.start_x) == ref["start"] and qa.tolist() == ref["qa"].tolist()
          and ta.tolist() == ref["ta"].tolist() and (D == ref["D"]).all())
    routed += 1 if res.flags & 4 else 0
    multi += 1 if (res.passes & 0x7f) >= 2 else 0
    serial += 1 if res.passes & 0x80 else 0
    if not ok or (N * M >= 16384 and not (res.flags & 4)):
        bad += 1
        print("MISMATCH", c, "sem", sem, mode, "N", N, "M", M, "gaps", dele, ext, "zero_rich", zero_rich, "flags", res.flags,
              "passes", hex(res.passes), flush=True)
    if c % 50 == 49:
        print("case", c + 1, "mismatches", bad, flush=True)
print("fuzz_wgpipe: %d cases (%d on the route, %d with several passes, %d ended in the strict-order routine), %d mismatches" % (cases, routed, multi, serial, bad))
sys.exit(1 if bad else 0)
