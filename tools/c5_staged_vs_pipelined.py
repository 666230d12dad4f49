import sys
sys.path.insert(0, '.')
import numpy as np
from aligner_amd import _ffi, workloads
from aligner_amd.batch import align_batch, align_batch_staged
from aligner_amd.matrices import get_blosum62
b = workloads.c5_batch(100000)
S = get_blosum62()
r = align_batch(b, _ffi.CORE_LOCAL, 11, 2, S).results
a = align_batch_staged(b, _ffi.CORE_LOCAL, 11, 2, S).results
a2 = align_batch_staged(b, _ffi.CORE_LOCAL, 11, 2, S).results
for f in r.dtype.names:
    d = np.nonzero(r[f] != a[f])[0]
    d2 = np.nonzero(a2[f] != a[f])[0]
    print(f, "pipelined vs staged differ:", len(d), "staged vs staged:", len(d2))
    for i in d[:6]:
        print("   pair", i, "N", int(b.q_len[i]), "M", int(b.t_len[i]), "pipelined", r[f][i] if f not in ("passes", "flags") else hex(int(r[f][i])), "staged", a[f][i] if f not in ("passes", "flags") else hex(int(a[f][i])))
