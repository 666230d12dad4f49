"""Histogram of aln_pair_result.passes over a C5 shard: how many pairs took the row-1 repair / a second full pass, and how
the cells of the re-filled pairs are distributed (they set the tail of a small shard).
usage: python tools/passes_hist.py [pairs=100000] [world=8] [del=11 ext=2 scale=1]   (e.g. 20000 1 46 9 2: BLOSUM62 x 2 with 46 / 9, the
integer form of the dyadic scheme x 0.5, 11.5 / 2.25)"""
import sys, numpy as np
sys.path.insert(0,'.')
from aligner_amd import _ffi, workloads
from aligner_amd.batch import StagedBatch
from aligner_amd.distributed import lpt_shards
from aligner_amd.matrices import get_blosum62
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ql, tl = workloads.c5_lengths(pairs)
b = workloads.c5_batch(pairs, indices=lpt_shards(ql * tl, world)[0])
# passes: see include/aligner_hip.h (bits 0-6 full passes, 8-15 repairs, 16-19: 1 + checkpoint where the repair stopped)
de = float(sys.argv[3]) if len(sys.argv) > 3 else 11.0
ex = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
sb = StagedBatch(b, _ffi.CORE_LOCAL, de, ex, get_blosum62() * scale, outputs=3)
sb.run(); sb.sync()
r = sb.fetch(False).results
vals, cnt = np.unique(r["passes"], return_counts=True)
print({hex(int(v)): int(c) for v, c in zip(vals, cnt)})
full = r["passes"] & 0x7f
cells = b.q_len.astype(np.int64) * b.t_len
for f in np.unique(full):
    sel = full == f
    print("full passes %d: %d pairs, cells max %.2e mean %.2e" % (f, sel.sum(), cells[sel].max(), cells[sel].mean()))
why = (r["passes"] >> 20) & 0xf
for v in np.unique(why):
    sel = why == v
    print("escalation reason %d: %d pairs, cells mean %.2e, single-strip %d" % (v, sel.sum(), cells[sel].mean(), (b.t_len[sel] <= 512).sum()))
print("flags", np.unique(r["flags"], return_counts=True))
