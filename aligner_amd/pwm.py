"""PWMAligner -- local alignment of a sequence against a position-weight matrix, on the GPU.

Mirror of aligner-core/src/pwm/mod.rs:9-126 (`PWMAligner<T>`, `AlignerTrait<T, PWMAlignment<T>>`) and of
`PWMAlignment` (aligner-core/src/alignment.rs:45-92).  It is the inner loop of the reference's latent-repeat-search
(bin/latent-repeat-search/engine/calc.rs:60-63, :121-124): one PWM, many windows -- `align_windows` below is that batch.
Same recurrence as the core "local" aligner (loop-carried penalty, no clamp, first row-major maximum) with the score of
cell (y, x) taken from matrix[[query[y-1], x-1]]; the traceback has no seed pair and an all-non-positive matrix yields an
empty alignment instead of a panic.
"""
import ctypes as C

import numpy as np

from . import _ffi
from . import runtime
from .alignment import AlignmentResult
from .batch import RESULT_DTYPE
from .enums import DNA


class PWMAlignment:
    """alignment.rs:45-52: numbered (PWM column or 0), query (residue or Blank), dim, coords, f."""

    def __init__(self, alphabet, numbered, query, dim, coords, f):
        self.alphabet = alphabet
        self.numbered = np.asarray(numbered, dtype=np.int64)
        self.query = np.asarray(query, dtype=np.uint8)
        self.dim = int(dim)
        self.coords = coords
        self.f = float(f)

    @classmethod
    def empty(cls, alphabet=DNA):
        """alignment.rs:82-91."""
        return cls(alphabet, [], [], 0, ((0, 0), (0, 0)), 0.0)

    def get_frequency_matrix(self):
        """alignment.rs:55-65: counts[[residue, column-1]] over columns that are neither a gap (0) nor Blank."""
        out = np.zeros((self.alphabet.volume(), self.dim), dtype=np.float64)
        keep = (self.numbered != 0) & (self.query != self.alphabet.blank())
        np.add.at(out, (self.query[keep].astype(np.int64), self.numbered[keep] - 1), 1.0)
        return out

    def get_alignment(self, _matrix=None):
        """alignment.rs:67-79: the residue where a PWM column is matched, Blank where it is not."""
        return np.where(self.numbered != 0, self.query, self.alphabet.blank()).astype(np.uint8)


class PWMAligner:
    """pwm/mod.rs:9-11, impl :13-126.  `from_seqs(query, _target)`: the second sequence is ignored, as in the reference."""

    def __init__(self, query, alphabet=DNA):
        self.alphabet = alphabet
        self.query = np.array(query, dtype=np.uint8, copy=True)

    @classmethod
    def from_str_seqs(cls, query, _target="", alphabet=DNA):
        return cls(alphabet.str_to_vec(query), alphabet)

    @classmethod
    def from_seqs(cls, query, _target=None, alphabet=DNA):
        return cls(query, alphabet)

    def perform_alignment(self, del_, ext, matrix, heuristics=None, *, want_matrices=False, device=None, **kw):
        """perform_alignment(del, ext, &pwm, heuristics): Err(UnnecessaryArgument) with heuristics, Err(MatrixShapeError)
        unless the matrix has 4 rows (pwm/mod.rs:36-42)."""
        res, numbered, qal, D, H = runtime.align_pwm(self.query, del_, ext, matrix,
                                                     heuristics_present=heuristics is not None,
                                                     want_directions=want_matrices, want_h=want_matrices, device=device,
                                                     blank=self.alphabet.blank(), **kw)
        coords = ((res.start_x + 1, res.end_x + 1), (res.start_y + 1, res.end_y + 1))       # pwm/mod.rs:118-121
        aln = PWMAlignment(self.alphabet, numbered, qal, np.asarray(matrix).shape[1], coords, res.f)
        return AlignmentResult(aln, alignment_matrix=H, direction_matrix=D, matrix=None, score=res.score,
                               summary={k: getattr(res, k) for k, _ in res._fields_})


def align_windows(windows, del_, ext, matrix, device=None, want_traceback=True, alphabet=DNA):
    """The latent-repeat-search inner loop as ONE batch: every window against the same PWM
    (engine/calc.rs:107-136 spawns threads stepping over windows and aligns them one by one).

    windows: iterable of residue-code arrays.  Returns (results ndarray, [PWMAlignment | None per window])."""
    wins = [np.asarray(w, dtype=np.uint8) for w in windows]
    n = len(wins)
    t_len = np.array([len(w) for w in wins], dtype=np.uint64)
    t_off = np.zeros(n, dtype=np.uint64)
    if n > 1:
        t_off[1:] = np.cumsum(t_len)[:-1]
    seqs = np.concatenate(wins) if n else np.zeros(0, dtype=np.uint8)
    return align_window_offsets(seqs, t_off, t_len, del_, ext, matrix, device=device, want_traceback=want_traceback,
                                alphabet=alphabet)


def align_window_offsets(sequence, starts, lengths, del_, ext, matrix, device=None, want_traceback=True, alphabet=DNA,
                         want_alignments=True, reuse=None):
    """The same for windows given as (start, length) into ONE residue-code array -- how the engine walks a chromosome
    (engine/calc.rs:111-124: `sequence[j..j + window]`): overlapping windows are not copied, the array goes to the GPU once per
    chunk.  `reuse`: a dict that keeps the output buffers between calls (a fresh 300 MB buffer costs more in page faults than
    the call itself).  Returns (results ndarray, [PWMAlignment | None per window] or None)."""
    lib = _ffi.load()
    seqs = np.ascontiguousarray(sequence, dtype=np.uint8)
    t_off = np.ascontiguousarray(starts, dtype=np.uint64)
    t_len = np.ascontiguousarray(lengths, dtype=np.uint64)
    n = len(t_off)
    m = np.asarray(matrix, dtype=np.float64)
    W = m.shape[1]
    q_off = np.zeros(n, dtype=np.uint64)
    q_len = np.full(n, W, dtype=np.uint64)
    outs = _ffi.OUT_SCORE | (_ffi.OUT_TRACEBACK if want_traceback else 0)
    p, keep = runtime.make_params(_ffi.PWM_LOCAL, del_, ext, m, outputs=outs, blank=alphabet.blank())
    res = np.zeros(n, dtype=RESULT_DTYPE)
    cap = (t_len + np.uint64(W + 2))
    tb_sz = ((np.uint64(5) * cap + np.uint64(3)) // np.uint64(4)) * np.uint64(4)
    tb_off = np.zeros(n, dtype=np.uint64)
    if n > 1:
        tb_off[1:] = np.cumsum(tb_sz)[:-1]
    need = (int(tb_sz.sum()) if want_traceback else 0) + 8
    tb = reuse.get("tb") if reuse is not None else None
    if tb is None or len(tb) < need:
        tb = np.zeros(need, dtype=np.uint8)
        if reuse is not None:
            reuse["tb"] = tb
    st = lib.aln_align_batch(runtime.context(device), C.byref(p), seqs.ctypes.data, q_off.ctypes.data,
                             q_len.ctypes.data, t_off.ctypes.data, t_len.ctypes.data, n, res.ctypes.data,
                             tb.ctypes.data if want_traceback else None, tb_off.ctypes.data if want_traceback else None)
    runtime.raise_for_status(st, "aln_align_batch(PWM)")
    if not want_alignments or not want_traceback:
        return res, None if not want_alignments else [None] * n
    alns = []
    for i in range(n):
        if not want_traceback or res["status"][i] != 0:
            alns.append(None)
            continue
        L, o, c = int(res["aln_len"][i]), int(tb_off[i]), int(cap[i])
        numbered = tb[o:o + 4 * L].view(np.uint32).copy()
        qal = tb[o + 4 * c:o + 4 * c + L].copy()
        r = res[i]
        coords = ((int(r["start_x"]) + 1, int(r["end_x"]) + 1), (int(r["start_y"]) + 1, int(r["end_y"]) + 1))
        alns.append(PWMAlignment(alphabet, numbered, qal, W, coords, float(r["f"])))
    return res, alns
