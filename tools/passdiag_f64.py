"""Which pairs of a real-valued batch take a different number of passes on the lean f64 strip and on run_strip (ALN_F64_OLD=1)?"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    os.environ.setdefault("ALN_NO_DYADIC", "1")
    from aligner_amd import _ffi, workloads
    from aligner_amd.batch import align_batch
    from aligner_amd.matrices import get_blosum62
    b = workloads.c5_batch(3000)
    r = align_batch(b, _ffi.CORE_LOCAL, 11.3, 2.1, get_blosum62() * 0.37, want_traceback=False)
    np.save(sys.argv[1], np.stack([r.results["passes"] & 0xff, b.q_len, b.t_len, r.results["score"].astype(np.int64)]))
else:
    def run(tag, **kw):
        env = dict(os.environ); env.update(kw); subprocess.check_call([sys.executable, __file__, "/tmp/p_%s.npy" % tag], env=env)
        return np.load("/tmp/p_%s.npy" % tag)
    o = run("old", ALN_F64_OLD="1")
    for tag, kw in (("new", {}), ("new_768_workgroups", {"ALN_FILL_WGS": "768"})):
        a = run(tag, **kw)
        d = np.nonzero(a[0] != o[0])[0]
        print(tag, "differ", len(d), "of", a.shape[1], "score equal", (a[3] == o[3]).all(), "first", d[:12])
