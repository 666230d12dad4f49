"""One PWM window at a time (PWMAligner / HeuristicPWMAligner's call): 330 nt against a 300-column PWM, real-valued weights
(f64 kernels) and integer ones.  usage: python tools/bench_pwm_single.py   (env ALN_NO_WGPIPE=1: one wave per pair)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from aligner_amd.pwm import PWMAligner
rng = np.random.default_rng(5)
seq = rng.integers(0, 4, 330).astype(np.uint8)
for name, pwm, de in (("real-valued", np.round(rng.normal(0, 1, (4, 300)), 3), (3.5, 1.25)), ("integer", rng.integers(-1, 2, (4, 300)).astype(np.float64), (3, 1))):
    for i in range(3):
        PWMAligner.from_seqs(seq).perform_alignment(de[0], de[1], pwm)
    ts = []
    for i in range(20):
        t0 = time.perf_counter(); r = PWMAligner.from_seqs(seq).perform_alignment(de[0], de[1], pwm); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%s PWM, one window 330 x 300: median %.3f ms min %.3f ms  f %.3f" % (name, ts[10] * 1e3, ts[0] * 1e3, r.alignment.f))
