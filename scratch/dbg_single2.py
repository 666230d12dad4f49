import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import oracle as orc
from aligner_amd import _ffi, runtime
N,M=500,700
for gaps in ((2,1),(3,1),(1,2),(11,2)):
    rng = np.random.default_rng(N + 31 * M + gaps[0])
    q = rng.integers(0, 4, N).astype(np.uint8)
    t = rng.integers(0, 4, M).astype(np.uint8)
    S = np.where(np.eye(4) > 0, 1.0, -1.0)
    ref = orc.align(orc.CORE_LOCAL, q, t, gaps[0], gaps[1], S, want_matrices=True)
    for mp in (0, 1):
        kw = dict(max_passes=mp) if mp else {}
        res, qa, ta, D, H = runtime.align_pair(_ffi.CORE_LOCAL, q, t, gaps[0], gaps[1], S, want_directions=True, want_h=False, **kw)
        bad = np.argwhere(D != ref["D"])
        print(gaps, "max_passes", mp, "passes", hex(res.passes), "score", res.score, ref["score"], "end", (res.end_y, res.end_x), ref["end"], "nbad", len(bad), flush=True)
