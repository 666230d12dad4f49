"""Throughput of the PWM window batch (latent-repeat-search's inner loop): W windows of 330 nt against one
300-column PWM, fill + traceback, inputs staged per call (host buffers; PCIe-inclusive)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aligner_amd.pwm import align_windows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(5)
pwm = rng.integers(-1, 2, (4, 300)).astype(np.float64)
chrom = rng.integers(0, 4, 30 * n + 400).astype(np.uint8)
wins = [chrom[i * 30:i * 30 + 330] for i in range(n)]
align_windows(wins[:1000], 3, 1, pwm)
for tb in (True, False):
    t0 = time.perf_counter()
    res, _ = align_windows(wins, 3, 1, pwm, want_traceback=tb)
    dt = time.perf_counter() - t0
    print("windows", n, "traceback", tb, "wall_s", round(dt, 3), "GCUPS(end-to-end incl. H2D/D2H + host packing)",
          round(n * 330 * 300 / dt / 1e9, 1), "ok", int((res["status"] == 0).sum()))
