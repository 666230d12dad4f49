import sys, time
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
from aligner_amd import _ffi, runtime, workloads
from aligner_amd.batch import RESULT_DTYPE
from aligner_amd.matrices import get_blosum62
from aligner_amd.statistics import shuffled_scores
S = get_blosum62()
q, t = workloads.c2_pair(homolog=False)
qp, tp = q[:350], t[:350]
_, _, pb = shuffled_scores(qp, tp, 0.0, 11, 2, S, rng=np.random.default_rng(1), device=0)
lib = _ffi.load(); ctx = runtime.context(0)
p, keep = runtime.make_params(_ffi.CORE_LOCAL, 11, 2, S, outputs=_ffi.OUT_SCORE)
res = np.zeros(len(pb), dtype=RESULT_DTYPE)
ts = []
for i in range(8):
    t0 = time.perf_counter()
    st = lib.aln_align_batch(ctx, C.byref(p), pb.seqs.ctypes.data, pb.q_off.ctypes.data, pb.q_len.ctypes.data, pb.t_off.ctypes.data, pb.t_len.ctypes.data, len(pb), res.ctypes.data, None, None)
    ts.append(time.perf_counter() - t0)
print("p-value batch: %d pairs, best %.3f ms median %.3f" % (len(pb), min(ts) * 1e3, sorted(ts)[4] * 1e3))
