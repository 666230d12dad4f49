"""Generates tests/golden/c5_100k_digest.npz: the CPU oracle's results for the FULL BASELINE C5 batch (100 000 protein pairs,
U[200,2000], core local, BLOSUM62, 11/2), condensed to
  score[i] (int32), and per block of 1000 pairs a SHA-256 over the block's
  (score, end_y, end_x, start_y, start_x, aln_len) int32 records followed by both aligned strings of every pair.
The GPU test recomputes the same digests from the HIP path's output (tests/test_gpu_parity.py).  ~15 min on 8 cores:
    python tests/golden/make_c5_golden.py [n_threads]
The oracle restates the reference (oracle/aln_oracle.c cites file:line); nothing here reads /root/reference."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as orc                                  # noqa: E402
from aligner_amd import workloads                     # noqa: E402
from aligner_amd.matrices import get_blosum62         # noqa: E402

BLOCK = 1000


def block_digest(summ, strings):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(summ, dtype=np.int32).tobytes())
    for qa, ta in strings:
        h.update(np.ascontiguousarray(qa, dtype=np.uint8).tobytes())
        h.update(np.ascontiguousarray(ta, dtype=np.uint8).tobytes())
    return h.digest()


def main():
    n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
    n = 100000
    S = get_blosum62()
    scores = np.zeros(n, dtype=np.int32)
    digests = []
    t0 = time.time()
    for lo in range(0, n, BLOCK):
        idx = np.arange(lo, lo + BLOCK)
        b = workloads.c5_batch(n, indices=idx)
        ref, tb, tb_off = orc.align_batch(orc.CORE_LOCAL, b.seqs, b.q_off, b.q_len, b.t_off, b.t_len, 11, 2, S, n_threads=n_threads)
        summ = np.zeros((BLOCK, 6), dtype=np.int32)
        strings = []
        for i in range(BLOCK):
            r = ref[i]
            assert r.status == 0
            summ[i] = (int(r.score), r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len)
            cap = int(b.q_len[i] + b.t_len[i]) + 2
            o = int(tb_off[i])
            strings.append((tb[o:o + r.aln_len], tb[o + cap:o + cap + r.aln_len]))
        scores[lo:lo + BLOCK] = summ[:, 0]
        digests.append(np.frombuffer(block_digest(summ, strings), dtype=np.uint8))
        if (lo // BLOCK) % 5 == 0:
            print("block %d / %d, %.0f s" % (lo // BLOCK, n // BLOCK, time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "c5_100k_digest.npz"), scores=scores, digests=np.stack(digests), block=BLOCK)
    print("done in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
