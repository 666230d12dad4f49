"""Randomized differential run of the BATCH driver: random batches (mixed lengths incl. empty sequences and one long
pair, every semantics, integer and real-valued matrices, forced generic / f64 kernels, score-only mode) against the
CPU oracle -- status, score, end / start cells and both aligned strings of every pair.
usage: python tools/fuzz_batch.py [batches [seed]]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi
from aligner_amd.batch import PairBatch, align_batch
from aligner_amd.matrices import get_blosum62

args = [a for a in sys.argv[1:] if not a.startswith('-')]
batches = int(args[0]) if len(args) > 0 else 40
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 99)
B62 = get_blosum62()
S4 = np.where(np.eye(24) > 0, 1.0, -1.0)
bad = 0
for b in range(batches):
    sem = int(rng.integers(0, 4))
    mode = ["fast", "generic", "f64", "real"][int(rng.integers(0, 4))] if sem < 2 else ["fast", "generic"][int(rng.integers(0, 2))]
    zero_rich = rng.random() < 0.4
    A = 4 if zero_rich else 20
    S = (S4 if zero_rich else B62).copy()
    if mode == "real":
        S = np.round(S + rng.normal(0, 0.3, S.shape), 3)
    dele, ext = [(2, 1), (1, 2), (11, 2), (3, 3), (11, 1)][int(rng.integers(0, 5))]
    if sem >= 2:
        ext = dele
    n = int(rng.integers(1, 120))
    pairs = []
    for i in range(n):
        u = rng.random()
        N = 0 if u < 0.02 else int(rng.integers(1, 64)) if u < 0.3 else int(rng.integers(64, 900)) if u < 0.97 else int(rng.integers(2000, 3000))
        M = 0 if rng.random() < 0.02 else int(rng.integers(1, 1300))
        q = rng.integers(0, A, N).astype(np.uint8)
        t = rng.integers(0, A, M).astype(np.uint8)
        if rng.random() < 0.4 and min(N, M) > 8:
            L = min(N, M) // 2
            t[:L] = q[:L]
        pairs.append((q, t))
    pb = PairBatch.from_pairs(pairs)
    want_tb = rng.random() < 0.8
    kw = dict(force_generic=True) if mode == "generic" else dict(force_f64=True) if mode == "f64" else {}
    print("batch", b, "sem", sem, mode, "pairs", n, "gaps", dele, ext, "zero_rich", zero_rich, "traceback", want_tb, flush=True)
    got = align_batch(pb, sem, dele, ext, S, want_traceback=want_tb, **kw)
    for i, (q, t) in enumerate(pairs):
        ref = orc.align(sem, q, t, dele, ext, S)
        r = got.results[i]
        ok = int(r["status"]) == ref["status"]
        if ok and ref["status"] == 0:
            ok = float(r["score"]) == ref["score"] and float(r["f"]) == ref["f"] and (int(r["end_y"]), int(r["end_x"])) == ref["end"]
            if ok and want_tb:
                qa, ta = got.aligned(i)
                ok = ((int(r["start_y"]), int(r["start_x"])) == ref["start"] and qa.tolist() == ref["qa"].tolist()
                      and ta.tolist() == ref["ta"].tolist())
        if not ok:
            bad += 1
            print("MISMATCH batch", b, "pair", i, "N", len(q), "M", len(t), "status", int(r["status"]), ref["status"],
                  "score", float(r["score"]), ref["score"], "flags", int(r["flags"]), "passes", hex(int(r["passes"])), flush=True)
print("done:", batches, "batches,", bad, "mismatches")
sys.exit(1 if bad else 0)
