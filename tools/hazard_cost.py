"""What the row-1 hazard machinery costs on C5: the same batch with del != ext (hazard: advice, checkpoints, repairs)
and with del == ext (no hazard: one plain pass) -- fill kernel time from the library's HIP events.
usage: python tools/hazard_cost.py [del ext]   (one setting: for a run under rocprofv3 --pmc SQ_INSTS_VALU)"""
import sys
sys.path.insert(0, '.')
from aligner_amd import _ffi, workloads
from aligner_amd.batch import StagedBatch
from aligner_amd.matrices import get_blosum62
b = workloads.c5_batch(40000)
sets = ((11, 2), (2, 2), (11, 11)) if len(sys.argv) < 3 else ((int(sys.argv[1]), int(sys.argv[2])),)
for de in sets:
    sb = StagedBatch(b, _ffi.CORE_LOCAL, de[0], de[1], get_blosum62(), outputs=3)
    sb.run(); sb.sync(); sb.enable_timing(True)
    for _ in range(3):
        sb.run()
    sb.sync()
    tm = sb.timing()
    print("del/ext %s: fill %.3f ms  traceback %.3f ms  fill GCUPS %.1f" % (de, tm["fill_ms"], tm["traceback_ms"], b.cells / tm["fill_ms"] / 1e6))
    sb.close()
