"""Large single pairs through the strip-pipelined route: how often does the row-1 advice change (a second run of the pipeline)?
Uniform-random and homolog 10k x 10k pairs over several seeds: fill / traceback time, passes, GCUPS end to end."""
import sys, numpy as np
sys.path.insert(0, '.')
from aligner_amd import _ffi, workloads
from aligner_amd.batch import PairBatch, StagedBatch
from aligner_amd.matrices import get_blosum62
S = get_blosum62()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for kind in ("uniform", "homolog"):
    for seed in range(6):
        q = workloads.random_codes(workloads.SEED_C4 + 31 * seed, n, 20)
        t = workloads.mutate(q, seed + 5, 20, 0.10, 0.02) if kind == "homolog" else workloads.random_codes(77 + seed, n, 20)
        one = PairBatch.from_pairs([(q, t)])
        sp = StagedBatch(one, _ffi.CORE_LOCAL, 11, 2, S, outputs=3)
        sp.run(); sp.sync()
        sp.enable_timing(True)
        for _ in range(3): sp.run()
        sp.sync()
        tm = sp.timing(); r = sp.fetch(False).results[0]
        tot = tm["fill_ms"] + tm["traceback_ms"]
        print("%s seed %d: %d x %d fill %.3f ms tb %.3f ms passes %d  %.1f GCUPS end to end  score %d" % (
            kind, seed, len(q), len(t), tm["fill_ms"], tm["traceback_ms"], int(r["passes"]) & 0x7f, one.cells / tot / 1e6, r["score"]), flush=True)
        sp.close()
