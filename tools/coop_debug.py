"""Debug aid for the cooperative passes: one C5-like batch against the oracle, mismatching pairs listed with shape and pass word.
usage: python tools/coop_debug.py [pairs lo hi]   (env ALN_NO_COOP / ALN_COOP_LINGER select the variant)"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd import _ffi, workloads
from aligner_amd.batch import align_batch
from aligner_amd.matrices import get_blosum62

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 700
S = get_blosum62()
b = workloads.c5_batch(n_pairs=n, lo=lo, hi=hi)
got = align_batch(b, _ffi.CORE_LOCAL, 11, 2, S)
ref, tb, tb_off = orc.align_batch(orc.CORE_LOCAL, b.seqs, b.q_off, b.q_len, b.t_off, b.t_len, 11, 2, S, 8)
bad = 0
hist = {}
for i in range(len(b)):
    r, g = ref[i], got.results[i]
    hist[hex(int(g["passes"]) & 0xf000ff)] = hist.get(hex(int(g["passes"]) & 0xf000ff), 0) + 1
    ok = int(g["status"]) == r.status
    if ok and r.status == 0:
        ok = (g["score"], g["end_y"], g["end_x"], g["start_y"], g["start_x"], g["aln_len"]) == (r.score, r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len)
        if ok:
            cap = int(b.q_len[i] + b.t_len[i]) + 2
            o = int(tb_off[i])
            qa, ta = got.aligned(i)
            ok = (qa == tb[o:o + r.aln_len]).all() and (ta == tb[o + cap:o + cap + r.aln_len]).all()
    if not ok:
        bad += 1
        if bad <= 20:
            print("MISMATCH pair", i, "N", int(b.q_len[i]), "M", int(b.t_len[i]), "status", int(g["status"]), r.status, "score", float(g["score"]), r.score,
                  "end", int(g["end_y"]), int(g["end_x"]), r.end_y, r.end_x, "start", int(g["start_y"]), int(g["start_x"]), r.start_y, r.start_x,
                  "len", int(g["aln_len"]), r.aln_len, "passes", hex(int(g["passes"])), flush=True)
print("pairs", len(b), "mismatches", bad, "pass words", hist)
