"""Fault injection for the single-pair route: one strip never runs (ALN_TEST_DROP_STRIP); the run must come back poisoned
(ERR_DEVICE) within the polls' bounds instead of hanging.  usage: ALN_TEST_DROP_STRIP=<s+1> python tools/abort_check.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from aligner_amd import _ffi, runtime
from aligner_amd.errors import DeviceError
from aligner_amd.matrices import get_blosum62
rng = np.random.default_rng(1)
q = rng.integers(0, 20, 2000).astype(np.uint8); t = rng.integers(0, 20, 2000).astype(np.uint8)
t0 = time.time()
try:
    res, qa, ta, D, H = runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, get_blosum62())
    print("returned status", res.status, "in %.2f s" % (time.time() - t0))
except DeviceError as e:
    print("DeviceError (expected):", e, "in %.2f s" % (time.time() - t0))
# the context must still be usable afterwards
import os
os.environ.pop("ALN_TEST_DROP_STRIP", None)
res, qa, ta, D, H = runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, get_blosum62())
print("next call status", res.status, "score", res.score)
