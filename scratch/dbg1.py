import numpy as np, sys
sys.path.insert(0,'.')
import oracle
from aligner_amd import _ffi, runtime
from aligner_amd.matrices import get_blosum62
S=get_blosum62()
N,M=9,1
rng=np.random.default_rng(N*1000+M)
q=rng.integers(0,20,N).astype(np.uint8); t=rng.integers(0,20,M).astype(np.uint8)
ref=oracle.align(1,q,t,11,2,S,want_matrices=True)
print("ref H", ref["H"][1]); print("ref D", ref["D"][1])
for kw in ({}, {"force_generic":True}, {"max_passes":1}, {"max_passes":20}):
    res,qa,ta,D,H=runtime.align_pair(1,q,t,11,2,S,want_directions=True,**kw)
    print(kw, "passes",hex(res.passes),"flags",res.flags,"D",D[1], "score",res.score)
