"""How long a wave of the batch fill works on a pair and how long it spends between two pairs (queue, descriptor, code check, profile
staging, summary) -- per-pair time stamps of a score-only run.  Needs the instrumented library:
    ALN_CXXFLAGS=-DALN_STAMPS python -m aligner_amd.build --force
usage: python tools/pair_gaps.py [c3|pwm|pvalue]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from aligner_amd import _ffi, workloads
from aligner_amd.batch import StagedBatch
from aligner_amd.matrices import get_blosum62, nucleotide_matrix
what = sys.argv[1] if len(sys.argv) > 1 else "c3"
if what == "c3":
    b = workloads.c3_batch(10000); sem, de, ex, S = _ffi.CORE_GLOBAL, 10, 1, nucleotide_matrix()
else:
    b = workloads.c5_batch(n_pairs=5000, lo=344, hi=350); sem, de, ex, S = _ffi.CORE_LOCAL, 11, 2, get_blosum62()
sb = StagedBatch(b, sem, de, ex, S, device=0, outputs=_ffi.OUT_SCORE)
sb.run(); sb.sync(); sb.run(); sb.sync()
r = sb.fetch(want_traceback=False).results
t0 = r["aln_len"].astype(np.int64); t1 = r["start_x"].astype(np.int64); wave = r["start_y"].astype(np.int64)
if (t0 == 0).all():
    raise SystemExit("no stamps: build with ALN_CXXFLAGS=-DALN_STAMPS")
base = t0.min()
s = (t0 - base) / 100.0; e = ((t1 - base) % (1 << 32)) / 100.0
print("%s: %d pairs, last end %.1f us; waves used %d" % (what, len(r), e.max(), len(np.unique(wave))))
dur = e - s
gaps = []
firsts = []
for w in np.unique(wave):
    idx = np.nonzero(wave == w)[0]
    o = idx[np.argsort(s[idx])]
    firsts.append(s[o[0]])
    gaps.extend((s[o[1:]] - e[o[:-1]]).tolist())
gaps = np.array(gaps)
print("in a pair (stamp to stamp: after the code check, up to the summary): mean %.1f us, median %.1f, p90 %.1f" % (dur.mean(), np.median(dur), np.percentile(dur, 90)))
print("between two pairs of a wave: mean %.1f us, median %.1f, p90 %.1f (%d gaps)" % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90), len(gaps)))
print("first pair of a wave starts at: mean %.1f us, max %.1f" % (np.mean(firsts), np.max(firsts)))
print("pairs per wave: mean %.2f max %d" % (len(r) / len(np.unique(wave)), np.bincount(wave.astype(np.int64)).max()))
