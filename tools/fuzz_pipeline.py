"""Randomized differential run of the PIPELINED batch call (aln_align_batch, aln_host.hip): random batches cut into random
numbers of chunks (ALN_CHUNK_CELLS), random fill depth, packed / sparse / shared-query sequence layouts, cumulative /
shuffled-with-gaps tb layouts, pairs the reference panics on (empty sequences, residue codes outside the matrix), every
semantics, integer and real-valued matrices, score-only mode, one device or a context naming device 0 two or three times --
against the CPU oracle: status, score, end / start cells and both aligned strings of every pair.
usage: python tools/fuzz_pipeline.py [batches [seed]]"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi
from aligner_amd.batch import PairBatch, align_batch
from aligner_amd.matrices import get_blosum62

args = [a for a in sys.argv[1:] if not a.startswith('-')]
batches = int(args[0]) if len(args) > 0 else 40
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 4242)
B62 = get_blosum62()
S4 = np.where(np.eye(24) > 0, 1.0, -1.0)
bad = 0
pairs_total = 0
for b in range(batches):
    sem = int(rng.integers(0, 4))
    mode = ["fast", "fast", "generic", "real"][int(rng.integers(0, 4))] if sem < 2 else ["fast", "generic"][int(rng.integers(0, 2))]
    zero_rich = rng.random() < 0.3
    A = 4 if zero_rich else 20
    S = (S4 if zero_rich else B62).copy()
    if mode == "real":
        S = np.round(S + rng.normal(0, 0.3, S.shape), 3)
    dele, ext = [(2, 1), (1, 2), (11, 2), (3, 3), (11, 1)][int(rng.integers(0, 5))]
    if sem >= 2:
        ext = dele
    n = int(rng.integers(2, 700))
    hi = int(rng.choice([60, 200, 500, 900]))
    pairs = []
    shared_q = rng.integers(0, A, int(rng.integers(5, hi))).astype(np.uint8) if rng.random() < 0.25 else None
    for i in range(n):
        N = 0 if rng.random() < 0.01 else int(rng.integers(1, hi))
        M = 0 if rng.random() < 0.01 else int(rng.integers(1, hi))
        q = shared_q if shared_q is not None else rng.integers(0, A, N).astype(np.uint8)
        t = rng.integers(0, A, M).astype(np.uint8)
        if rng.random() < 0.3 and min(len(q), M) > 8:
            L = min(len(q), M) // 2
            t[:L] = q[:L]
        if rng.random() < 0.01 and M > 0:
            t = t.copy(); t[int(rng.integers(0, M))] = 24 + int(rng.integers(0, 200))       # outside the 24 x 24 matrix
        pairs.append((np.array(q), t))
    # sequence layout: packed, shared query at one offset, or sparse
    layout = "shared" if shared_q is not None else ["packed", "sparse"][int(rng.random() < 0.3)]
    if layout == "packed":
        pb = PairBatch.from_pairs(pairs)
    elif layout == "shared":
        seqs = [shared_q]; pos = len(shared_q); q_off, q_len, t_off, t_len = [], [], [], []
        for q, t in pairs:
            q_off.append(0); q_len.append(len(shared_q)); t_off.append(pos); t_len.append(len(t)); seqs.append(t); pos += len(t)
        pb = PairBatch(np.concatenate(seqs), q_off, q_len, t_off, t_len)
    else:
        gap = 4000
        big = np.full(2 * n * gap + 8, 9, dtype=np.uint8)
        q_off = (np.arange(n, dtype=np.uint64) * 2 + 1) * gap
        t_off = (np.arange(n, dtype=np.uint64) * 2) * gap + 3
        for i, (q, t) in enumerate(pairs):
            big[int(q_off[i]):int(q_off[i]) + len(q)] = q
            big[int(t_off[i]):int(t_off[i]) + len(t)] = t
        pb = PairBatch(big, q_off, [len(q) for q, _ in pairs], t_off, [len(t) for _, t in pairs])
    want_tb = rng.random() < 0.8
    tb_off = None
    if want_tb and rng.random() < 0.35:                       # foreign layout: shuffled order, gaps
        cap = 2 * (pb.q_len + pb.t_len + np.uint64(2)) + np.uint64(int(rng.integers(0, 9)))
        order = rng.permutation(n)
        tb_off = np.zeros(n, dtype=np.uint64)
        tb_off[order[1:]] = np.cumsum(cap[order])[:-1]
    chunks = int(rng.choice([1, 2, 3, 5, 9, 17, 40]))
    os.environ["ALN_CHUNK_CELLS"] = str(max(1, pb.cells // chunks))
    os.environ["ALN_FILL_DEPTH"] = str(int(rng.integers(1, 4)))
    devices = [None, None, [0, 0], [0, 0, 0]][int(rng.integers(0, 4))]
    kw = dict(force_generic=True) if mode == "generic" else {}
    print("batch", b, "sem", sem, mode, "pairs", n, "<", hi, "gaps", dele, ext, "zero_rich", zero_rich, "tb", want_tb,
          "foreign" if tb_off is not None else "std", layout, "chunks~", chunks, "devices", devices, flush=True)
    got = align_batch(pb, sem, dele, ext, S, want_traceback=want_tb, tb_off=tb_off, devices=devices, **kw)
    ref, rtb, roff = orc.align_batch(sem, pb.seqs, pb.q_off, pb.q_len, pb.t_off, pb.t_len, dele, ext, S, n_threads=8)
    pairs_total += n
    for i in range(n):
        r, g = ref[i], got.results[i]
        ok = int(g["status"]) == r.status
        if ok and r.status == 0:
            ok = (float(g["score"]), float(g["f"]), int(g["end_y"]), int(g["end_x"])) == (r.score, r.f, r.end_y, r.end_x)
            if ok and want_tb:
                cap = int(pb.q_len[i] + pb.t_len[i]) + 2
                o = int(roff[i])
                qa, ta = got.aligned(i)
                ok = ((int(g["start_y"]), int(g["start_x"]), int(g["aln_len"])) == (r.start_y, r.start_x, r.aln_len)
                      and (qa == rtb[o:o + r.aln_len]).all() and (ta == rtb[o + cap:o + cap + r.aln_len]).all())
        if not ok:
            bad += 1
            print("MISMATCH batch", b, "pair", i, "N", int(pb.q_len[i]), "M", int(pb.t_len[i]), "status", int(g["status"]), r.status,
                  "score", float(g["score"]), r.score, "flags", int(g["flags"]), "passes", hex(int(g["passes"])), flush=True)
print("fuzz_pipeline: %d batches, %d pairs, %d mismatches" % (batches, pairs_total, bad))
sys.exit(1 if bad else 0)
