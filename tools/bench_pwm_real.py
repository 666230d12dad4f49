import sys, time
import numpy as np
sys.path.insert(0, ".")
from aligner_amd.pwm import align_windows
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(5)
chrom = rng.integers(0, 4, 30 * n + 400).astype(np.uint8)
wins = [chrom[i * 30:i * 30 + 330] for i in range(n)]
for name, pwm, de in (("integer PWM", rng.integers(-1, 2, (4, 300)).astype(np.float64), (3, 1)),
                      ("real-valued PWM", np.round(rng.normal(0, 1, (4, 300)), 3), (3.5, 1.25))):
    align_windows(wins[:1000], de[0], de[1], pwm)
    for tb in (False, True):
        t0 = time.perf_counter()
        res, _ = align_windows(wins, de[0], de[1], pwm, want_traceback=tb)
        dt = time.perf_counter() - t0
        print(name, "windows", n, "traceback", tb, "wall_s %.3f" % dt, "GCUPS %.1f" % (n * 330 * 300 / dt / 1e9), "ok", int((res["status"] == 0).sum()), flush=True)
