import numpy as np, sys
sys.path.insert(0,'.')
import oracle
from aligner_amd import runtime
rng=np.random.default_rng(40*7+30)
pwm=rng.integers(-1,2,(4,30)).astype(np.float64)
seq=rng.integers(0,4,40).astype(np.uint8)
ref=oracle.align_pwm(seq,3,1,pwm)
res,numbered,qal,D,H=runtime.align_pwm(seq,3,1,pwm)
print("ref num",ref["numbered"]); print("gpu num",numbered)
print("ref qal",ref["qal"]); print("gpu qal",qal)
print(res.aln_len, res.start_y,res.start_x,res.end_y,res.end_x, ref["start"], ref["end"])
