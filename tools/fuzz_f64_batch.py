"""Randomized differential run of the real-valued BATCH path (aln_fill_f64_kernel, the lean f64 strip) against the CPU oracle: random
batch sizes, lengths up to three strips, both core semantics, real-valued matrices (scaled BLOSUM62, random normal weights, a
zero-rich +-0.5 scheme kept on the f64 kernels), gap costs with del = ext and del != ext, related and unrelated pairs -- summaries
and both aligned strings of every pair.  usage: python tools/fuzz_f64_batch.py [batches [seed]]"""
import os
os.environ.setdefault("ALN_NO_DYADIC", "1")      # dyadic schemes stay on the f64 kernels: they are the row-1 hazard's worst case
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi
from aligner_amd.batch import PairBatch, align_batch
from aligner_amd.matrices import get_blosum62

args = [a for a in sys.argv[1:] if not a.startswith('-')]
batches = int(args[0]) if len(args) > 0 else 100
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 64)
B62 = get_blosum62()
bad = pairs_done = 0
for b in range(batches):
    sem = [_ffi.CORE_LOCAL, _ffi.CORE_GLOBAL][int(rng.integers(0, 2))]
    kind = int(rng.integers(0, 3))
    if kind == 0:
        A, S = 20, B62 * float(rng.choice([0.37, 0.5, 1.0 / 3.0]))
    elif kind == 1:
        A = int(rng.integers(2, 25)); S = np.round(rng.normal(0, 2, (A, A)), 2); S = (S + S.T) / 2 + np.eye(A) * 3.1
    else:
        A, S = 4, np.where(np.eye(4) > 0, 0.5, -0.5)
    dele, ext = [(2.5, 1.25), (1.1, 2.3), (11.3, 2.1), (3.3, 3.3), (0.5, 0.5), (1.0, 0.5)][int(rng.integers(0, 6))]
    n = int(rng.integers(5, 70))
    hi = [60, 400, 1500][int(rng.integers(0, 3))]
    pairs = []
    for _ in range(n):
        N, M = int(rng.integers(1, hi)), int(rng.integers(1, hi))
        q = rng.integers(0, A, N).astype(np.uint8); t = rng.integers(0, A, M).astype(np.uint8)
        if rng.random() < 0.5 and min(N, M) > 8:
            L = min(N, M) // 2
            t[M // 4:M // 4 + L] = q[N // 4:N // 4 + L][:len(t[M // 4:M // 4 + L])]
        pairs.append((q, t))
    pb = PairBatch.from_pairs(pairs)
    got = align_batch(pb, sem, dele, ext, S)
    ref, tb, tb_off = orc.align_batch(sem, pb.seqs, pb.q_off, pb.q_len, pb.t_off, pb.t_len, dele, ext, S, n_threads=8)
    for i in range(n):
        r, g = ref[i], got.results[i]
        ok = g["status"] == r.status
        if ok and r.status == 0:
            ok = (g["score"], g["f"], g["end_y"], g["end_x"], g["start_y"], g["start_x"], g["aln_len"]) == \
                 (r.score, r.f, r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len)
            if ok:
                cap = int(pb.q_len[i] + pb.t_len[i]) + 2; o = int(tb_off[i]); qa, ta = got.aligned(i)
                ok = bool((qa == tb[o:o + r.aln_len]).all() and (ta == tb[o + cap:o + cap + r.aln_len]).all())
        if not ok:
            bad += 1
            print("MISMATCH batch", b, "pair", i, "sem", sem, "kind", kind, "gaps", dele, ext, "N", int(pb.q_len[i]), "M", int(pb.t_len[i]),
                  "passes", hex(int(g["passes"])), "flags", int(g["flags"]), flush=True)
    pairs_done += n
    assert not (got.results["flags"] & 1).any(), "batch ran on the integer kernels"
    if b % 25 == 24:
        print("batch", b + 1, "pairs", pairs_done, "mismatches", bad, flush=True)
print("done:", batches, "batches,", pairs_done, "pairs,", bad, "mismatches")
sys.exit(1 if bad else 0)
