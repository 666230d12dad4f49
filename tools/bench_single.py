"""One large pair (default 10 000 x 10 000, C4) through the single-pair route: fill / traceback times from the
library's HIP events.  usage: python tools/bench_single.py [N [M]]   (env ALN_SINGLE_R, ALN_SINGLE_W1 select variants)"""
import sys, time, numpy as np
sys.path.insert(0,'.')
import torch
from aligner_amd import _ffi, workloads
from aligner_amd.batch import PairBatch, StagedBatch
from aligner_amd.matrices import get_blosum62
S=get_blosum62()
n=int(sys.argv[1]) if len(sys.argv)>1 else 10000
m=int(sys.argv[2]) if len(sys.argv)>2 else n
q,_=workloads.c4_pair(False,n=n)
_,t=workloads.c4_pair(False,n=m)
one=PairBatch.from_pairs([(q,t)])
sp=StagedBatch(one,_ffi.CORE_LOCAL,11,2,S,outputs=3)
sp.run(); sp.sync()
sp.enable_timing(True)
for _ in range(5): sp.run()
sp.sync()
tm=sp.timing(); r=sp.fetch(False).results[0]
print("N",n,"M",m,"fill_ms",round(tm["fill_ms"],3),"tb_ms",round(tm["traceback_ms"],3),"fill GCUPS",round(one.cells/tm["fill_ms"]/1e6,1),"passes",hex(r["passes"]),"score",r["score"])
