"""Result types of the hot path.

Host mirror of aligner-core/src/alignment.rs:4-43 (`Alignment<T>`: query, target, coords, f;
`get_frequency_matrix`, `get_alignment`) and aligner-core/src/alignment_result.rs:6-13 (`AlignmentResult`).
The aligned strings arrive from the GPU traceback kernel as residue-code bytes.
"""
import numpy as np


class Alignment:
    """alignment.rs:4-10.  `query` / `target` are uint8 code arrays of equal length (Blank = 98)."""

    def __init__(self, alphabet, query, target, coords, f):
        self.alphabet = alphabet
        self.query = np.asarray(query, dtype=np.uint8)
        self.target = np.asarray(target, dtype=np.uint8)
        self.coords = coords
        self.f = float(f)

    def get_frequency_matrix(self):
        """alignment.rs:13-23: counts[[y, x]] over columns where neither side is blank."""
        v = self.alphabet.volume()
        blank = self.alphabet.blank()
        keep = (self.query != blank) & (self.target != blank)
        out = np.zeros((v, v), dtype=np.float64)
        np.add.at(out, (self.target[keep].astype(np.int64), self.query[keep].astype(np.int64)), 1.0)
        return out

    def get_alignment(self, matrix):
        """alignment.rs:25-42: the midline -- residue if equal, Pos if both non-blank and S[y][x] >= 0, else Blank."""
        m = np.asarray(matrix, dtype=np.float64)
        blank, pos = self.alphabet.blank(), self.alphabet.pos()
        x, y = self.query, self.target
        both = (x != blank) & (y != blank)
        score_ok = np.zeros(len(x), dtype=bool)
        score_ok[both] = m[y[both].astype(np.int64), x[both].astype(np.int64)] >= 0.0
        out = np.where(x == y, x, np.where(score_ok, pos, blank)).astype(np.uint8)
        return out

    def query_str(self):
        return self.alphabet.vec_to_str(self.query)

    def target_str(self):
        return self.alphabet.vec_to_str(self.target)

    def midline_str(self, matrix):
        return self.alphabet.vec_to_str(self.get_alignment(matrix))

    def __repr__(self):
        return "Alignment { query: %s, target: %s, coords: %r, f: %r }" % (
            self.query_str(), self.target_str(), self.coords, self.f)


class AlignmentResult:
    """alignment_result.rs:6-13.  `alignment_matrix` (f64 H) and `direction_matrix` are only materialised when the
    caller asked for them (no code in the reference reads them after construction; they cost 9 B/cell)."""

    def __init__(self, alignment, alignment_matrix=None, direction_matrix=None, matrix=None, score=None, summary=None):
        self.alignment = alignment
        self.alignment_matrix = alignment_matrix
        self.direction_matrix = direction_matrix
        self.matrix = matrix
        self.score = score          # H[M][N] (global) / H max (local); the reference leaves f = 0 for core global
        self.summary = summary      # raw aln_pair_result fields
