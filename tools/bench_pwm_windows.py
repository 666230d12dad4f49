"""bench.py's PWM window batch on its own (for profiling): 100 000 windows of 330 nt, one every 30 nt of a chromosome, against a
4 x 300 PWM, del 3 / ext 1, score only, windows as (start, length) into the one chromosome array.
usage: python tools/bench_pwm_windows.py [calls=4] [windows=100000]   (env PWM_TB=1: with the strings)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from aligner_amd.pwm import align_window_offsets
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
tb = os.environ.get("PWM_TB") is not None
rng = np.random.default_rng(300)
pwm = rng.integers(-3, 4, (4, 300)).astype(np.float64)
chrom = rng.integers(0, 4, n * 30 + 400).astype(np.uint8)
starts, lens = np.arange(n, dtype=np.uint64) * np.uint64(30), np.full(n, 330, dtype=np.uint64)
keep = {}
align_window_offsets(chrom, starts, lens, 3, 1, pwm, want_traceback=tb, want_alignments=False, reuse=keep)
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    res, _ = align_window_offsets(chrom, starts, lens, 3, 1, pwm, want_traceback=tb, want_alignments=False, reuse=keep)
    ts.append(time.perf_counter() - t0)
print("%d windows: best %.2f ms median %.2f ms = %.1f GCUPS; repaired or re-filled %d, filled twice %d" % (
    n, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, n * 330 * 300 / min(ts) / 1e9, int(((res["passes"] >> 8) != 0).sum()), int(((res["passes"] & 0xff) >= 2).sum())))
