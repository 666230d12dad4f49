"""Who works on what, when: per-pair start / end stamps of the batch fill kernel for rank 0's shard of C5 at world size W.
Needs an instrumented library (time stamps in result fields the score-only run does not use):
    ALN_CXXFLAGS=-DALN_STAMPS python -m aligner_amd.build --force
usage: python tools/tail_timeline.py [W=8]   (env ALN_FILL_WGS=n: n workgroups instead of 3 per CU)
Prints when the queue ran dry, when the last pair ended, the pairs that ended last, and -- per "class" of wave (class c = the
c-th workgroup that became resident on its CU, i.e. the c-th oldest wave of its SIMD) -- how long the pairs taken at t = 0 took."""
import sys
sys.path.insert(0, '.')
import numpy as np
from aligner_amd import _ffi, workloads
from aligner_amd.batch import StagedBatch
from aligner_amd.distributed import lpt_shards
from aligner_amd.matrices import get_blosum62

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
qlen, tlen = workloads.c5_lengths(100000)
shards = lpt_shards(qlen * tlen, W)
batch = workloads.c5_batch(100000, indices=shards[0])
sb = StagedBatch(batch, _ffi.CORE_LOCAL, 11, 2, get_blosum62(), device=0, outputs=_ffi.OUT_SCORE)
sb.run(); sb.sync(); sb.run(); sb.sync()
r = sb.fetch(want_traceback=False).results
t0 = r["aln_len"].astype(np.int64); t1 = r["start_x"].astype(np.int64)
if (t0 == 0).all():
    raise SystemExit("no stamps: build with ALN_CXXFLAGS=-DALN_STAMPS")
base = t0.min()
s = (t0 - base) / 100.0; e = ((t1 - base) % (1 << 32)) / 100.0       # us (100 MHz)
cells = batch.q_len.astype(np.int64) * batch.t_len.astype(np.int64)
passes = r["passes"]; wave = r["start_y"].astype(np.int64)
print("pairs %d, %.3g cells; last pair taken at %.0f us, last pair done at %.0f us (ideal at 2.69 TCUPS: %.0f us)" % (
    len(r), cells.sum(), s.max(), e.max(), cells.sum() / 2.69e6))
for thr in (0.8, 0.9, 0.95):
    m = e > thr * e.max()
    print("  done after %.0f us: %d pairs, %d of them with a second full pass" % (thr * e.max(), m.sum(), ((passes[m] & 0xff) >= 2).sum()))
first = s < 50
for c in range(3):
    m = first & (wave // 1024 == c)
    m2 = m & ((passes & 0xff) >= 2)
    if m.any():
        print("pairs taken at t = 0 by class-%d waves: %d, mean %.3g cells, done at median %.0f us = %.2f GCUPS per wave; %d with a second pass, done at median %.0f max %.0f" % (
            c, m.sum(), cells[m].mean(), np.median(e[m]), np.median(cells[m] / e[m]) / 1e3, m2.sum(), np.median(e[m2]) if m2.any() else 0, e[m2].max() if m2.any() else 0))
for i in np.argsort(-e)[:12]:
    print("  pair %6d %.3g cells (%d x %d) taken %.0f done %.0f us passes %#x wave %d: %.2f GCUPS" % (
        i, cells[i], batch.q_len[i], batch.t_len[i], s[i], e[i], passes[i], wave[i], cells[i] * (passes[i] & 0xff) / (e[i] - s[i]) / 1e3))
for t in range(0, int(e.max()) + 1, 500):
    print("  t = %5d us: %d pairs in flight" % (t, ((s <= t) & (e > t)).sum()))
