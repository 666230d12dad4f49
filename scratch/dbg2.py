import numpy as np, sys
sys.path.insert(0,'.')
from aligner_amd import _ffi
_ffi.LIB_PATH='scratch/libdbg.so'
from aligner_amd import runtime
from aligner_amd.matrices import get_blosum62
S=get_blosum62()
N,M=9,1
rng=np.random.default_rng(N*1000+M)
q=rng.integers(0,20,N).astype(np.uint8); t=rng.integers(0,20,M).astype(np.uint8)
print(q,t, [S[t[0],c] for c in q])
res,qa,ta,D,H=runtime.align_pair(1,q,t,11,2,S,want_directions=True)
print(D[1])
