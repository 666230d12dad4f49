"""Randomized differential run of the PWM aligner (pwm/mod.rs) as window batches: random window lengths (incl. longer
than one tracker chunk), PWM widths, integer and real-valued weights, gap costs, vs the CPU oracle.
usage: python tools/fuzz_pwm.py [batches [seed]]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from aligner_amd.pwm import align_windows

args = [a for a in sys.argv[1:] if not a.startswith('-')]
batches = int(args[0]) if len(args) > 0 else 30
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 3)
bad = 0
for b in range(batches):
    W = int(rng.integers(1, 700)) if rng.random() < 0.9 else int(rng.integers(1500, 2000))
    real = rng.random() < 0.5
    pwm = np.round(rng.normal(0, 1.5, (4, W)), 2) if real else rng.integers(-3, 4, (4, W)).astype(np.float64)
    dele, ext = [(3, 1), (1, 2), (2, 2), (4, 1)][int(rng.integers(0, 4))]
    n = int(rng.integers(1, 60))
    wins = [rng.integers(0, 4, int(rng.integers(1, 900))).astype(np.uint8) for _ in range(n)]
    want_tb = rng.random() < 0.8
    print("batch", b, "W", W, "real" if real else "int", "windows", n, "gaps", dele, ext, "traceback", want_tb, flush=True)
    res, alns = align_windows(wins, dele, ext, pwm, want_traceback=want_tb)
    for i, w in enumerate(wins):
        ref = orc.align_pwm(w, dele, ext, pwm)
        r = res[i]
        ok = int(r["status"]) == ref["status"]
        if ok and ref["status"] == 0:
            ok = float(r["f"]) == ref["f"] and (int(r["end_y"]), int(r["end_x"])) == ref["end"]
            if ok and want_tb:
                a = alns[i]
                ok = a.coords == ref["coords"] and a.numbered.tolist() == ref["numbered"].tolist() and a.query.tolist() == ref["qal"].tolist()
        if not ok:
            bad += 1
            print("MISMATCH batch", b, "window", i, "len", len(w), "status", int(r["status"]), ref["status"], "f", float(r["f"]), ref["f"], flush=True)
print("done:", batches, "batches,", bad, "mismatches")
sys.exit(1 if bad else 0)
