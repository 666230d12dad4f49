"""Randomized differential run of the OVERLAPPED traceback (batches of >= 4096 pairs, where the walk kernel runs beside the
fill kernel): random short pairs, every semantics, zero-rich alphabets (row-1 hazard repairs, strict-order fallbacks whose
directions are stored with ordinary stores), against the CPU oracle -- summaries and both aligned strings of every pair.
usage: python tools/fuzz_overlap.py [batches [seed]]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi
from aligner_amd.batch import PairBatch, align_batch
from aligner_amd.matrices import get_blosum62

args = [a for a in sys.argv[1:] if not a.startswith('-')]
batches = int(args[0]) if len(args) > 0 else 8
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 7)
B62 = get_blosum62()
S4 = np.where(np.eye(24) > 0, 1.0, -1.0)
bad = 0
for b in range(batches):
    sem = 1 if rng.random() < 0.5 else int(rng.integers(0, 4))      # core local (hazards, repairs, fallbacks) half of the time
    zero_rich = rng.random() < 0.5
    A = 4 if zero_rich else 20
    S = S4 if zero_rich else B62
    dele, ext = [(2, 1), (1, 2), (11, 2), (3, 3), (11, 1)][int(rng.integers(0, 5))]
    if sem >= 2:
        ext = dele
    n = int(rng.integers(4100, 6500))
    hi = int(rng.integers(60, 700))
    pairs = []
    for i in range(n):
        N = int(rng.integers(1, hi))
        M = int(rng.integers(1, hi))
        q = rng.integers(0, A, N).astype(np.uint8)
        t = rng.integers(0, A, M).astype(np.uint8)
        if rng.random() < 0.4 and min(N, M) > 8:
            L = min(N, M) // 2
            t[:L] = q[:L]
        pairs.append((q, t))
    pb = PairBatch.from_pairs(pairs)
    kw = {"max_passes": 1} if (zero_rich and rng.random() < 0.5) else {}
    got = align_batch(pb, sem, dele, ext, S, **kw)
    ref, tb, tb_off = orc.align_batch(sem, pb.seqs, pb.q_off, pb.q_len, pb.t_off, pb.t_len, dele, ext, S, n_threads=16)
    nb = 0
    for i in range(n):
        r, g = ref[i], got.results[i]
        ok = g["status"] == r.status
        if ok and r.status == 0:
            ok = (g["score"], g["f"], g["end_y"], g["end_x"], g["start_y"], g["start_x"], g["aln_len"]) == \
                 (r.score, r.f, r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len)
            if ok:
                cap = int(pb.q_len[i] + pb.t_len[i]) + 2
                o = int(tb_off[i])
                qa, ta = got.aligned(i)
                ok = bool((qa == tb[o:o + r.aln_len]).all() and (ta == tb[o + cap:o + cap + r.aln_len]).all())
        nb += not ok
    serial = int((got.results["passes"] & 0x80 != 0).sum())
    print("batch %d: sem %d %s del %d ext %d n %d hi %d %s: %d mismatches (strict-order pairs %d)" % (
        b, sem, "zero-rich" if zero_rich else "blosum", dele, ext, n, hi, kw, nb, serial), flush=True)
    bad += nb
print("done: %d batches, %d mismatches" % (batches, bad))
sys.exit(1 if bad else 0)
