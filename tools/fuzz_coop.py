"""Randomized differential run of the cooperative passes of the fast batch kernel: batches of pairs with several strips (rows up to
~5000), few or many pairs per resident wave, core local (re-fills: zero-rich scoring makes them frequent) / core global / legacy,
score only or with strings, and the testing knobs of the machinery (ALN_COOP_DEBUG 0: as shipped, 2: every open pass filled by its
owner alone, 8: every re-fill shared, 10: both; ALN_COOP_TAIL: first passes opened or not; ALN_COOP_LINGER) -- every pair against
the CPU oracle (status, score, end / start cells, both aligned strings).
usage: python tools/fuzz_coop.py [batches [seed]]"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi
from aligner_amd.batch import PairBatch, align_batch
from aligner_amd.matrices import get_blosum62

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
B62 = get_blosum62()
S4 = np.where(np.eye(24) > 0, 1.0, -1.0)
bad = total = refills = 0
for b in range(batches):
    sem = [1, 1, 1, 0, 2, 3][int(rng.integers(0, 6))]
    zero_rich = sem == 1 and rng.random() < 0.4
    A = 4 if zero_rich else 20
    S = (S4 if zero_rich else B62)
    dele, ext = [(11, 2), (2, 1), (3, 1), (11, 1)][int(rng.integers(0, 4))]
    if sem >= 2:
        ext = dele
    # core semantics: every third batch as a dyadic real-valued scheme (the same scheme divided by 2, 4 or 8: the library scales it
    # back onto the integer kernels; the oracle computes in f64 on the numbers as given)
    dyadic = int(rng.integers(1, 4)) if sem < 2 and rng.random() < 0.34 else 0
    if dyadic:
        S = S / float(1 << dyadic); dele = dele / float(1 << dyadic); ext = ext / float(1 << dyadic)
    shape = int(rng.integers(0, 4))
    if shape == 0:      # few large pairs: fewer pairs than waves, everything shared
        n = int(rng.integers(2, 60)); lo, hi = 600, 5000
    elif shape == 1:    # a small batch of mixed pairs
        n = int(rng.integers(60, 800)); lo, hi = 30, 2500
    elif shape == 2:    # several pairs per wave
        n = int(rng.integers(3000, 9000)); lo, hi = 20, 900
    else:               # tall and narrow / short and wide
        n = int(rng.integers(100, 1500)); lo, hi = 16, 3000
    pairs = []
    for i in range(n):
        N = int(rng.integers(lo, hi)); M = int(rng.integers(lo, hi))
        if shape == 3:
            if rng.random() < 0.5: N = int(rng.integers(16, 200))
            else: M = int(rng.integers(16, 200))
        q = rng.integers(0, A, N).astype(np.uint8); t = rng.integers(0, A, M).astype(np.uint8)
        if rng.random() < 0.3 and min(N, M) > 8:
            L = min(N, M) // 2; t[:L] = q[:L]
        pairs.append((q, t))
    pb = PairBatch.from_pairs(pairs)
    dbg = [0, 0, 2, 8, 10][int(rng.integers(0, 5))]
    tail = [None, None, "0", "1000000"][int(rng.integers(0, 4))]
    linger = [None, "0", "1"][int(rng.integers(0, 3))]
    os.environ["ALN_COOP_DEBUG"] = str(dbg)
    for k, v in (("ALN_COOP_TAIL", tail), ("ALN_COOP_LINGER", linger)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    want_tb = rng.random() < 0.8
    print("batch", b, "sem", sem, "pairs", n, "shape", shape, "gaps", dele, ext, "zero_rich", zero_rich, "debug", dbg, "tail", tail, "linger", linger,
          "traceback", want_tb, "dyadic", dyadic, "cells %.3g" % pb.cells, flush=True)
    got = align_batch(pb, sem, dele, ext, S, want_traceback=want_tb)
    ref, tb, tb_off = orc.align_batch(sem, pb.seqs, pb.q_off, pb.q_len, pb.t_off, pb.t_len, dele, ext, S, 16)
    for i in range(n):
        r, g = ref[i], got.results[i]
        ok = int(g["status"]) == r.status
        if ok and r.status == 0:
            ok = (g["score"], g["f"], g["end_y"], g["end_x"]) == (r.score, r.f, r.end_y, r.end_x)
            if ok and want_tb:
                cap = int(pb.q_len[i] + pb.t_len[i]) + 2
                o = int(tb_off[i])
                qa, ta = got.aligned(i)
                ok = (g["start_y"], g["start_x"], g["aln_len"]) == (r.start_y, r.start_x, r.aln_len) and \
                    (qa == tb[o:o + r.aln_len]).all() and (ta == tb[o + cap:o + cap + r.aln_len]).all()
        total += 1
        refills += int((int(g["passes"]) & 0xff) >= 2)
        if not ok:
            bad += 1
            print("MISMATCH batch", b, "pair", i, "N", int(pb.q_len[i]), "M", int(pb.t_len[i]), "status", int(g["status"]), r.status, "score", float(g["score"]), r.score,
                  "passes", hex(int(g["passes"])), "flags", int(g["flags"]), flush=True)
print("fuzz_coop: %d batches, %d pairs (%d re-filled), %d mismatches" % (batches, total, refills, bad))
sys.exit(1 if bad else 0)
