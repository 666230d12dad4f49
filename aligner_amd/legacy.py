"""SimpleAligner -- the legacy i32 / linear-gap aligner, on the GPU.

Mirror of src/align/aligner_core.rs:72-84 (`SimpleAligner`, trait `Aligner`), :96-183 (`global_alignment`),
:185-269 (`local_alignment`) and the result structs :13-70.  This is the variant the reference's only
known-answer tests exercise (src/tests/test_alignment.rs).
"""
import numpy as np

from . import _ffi
from . import runtime
from .enums import Protein


class _LegacyResult:
    def __init__(self, alignment_matrix, direction_matrix, optimal_alignment, max_f=None):
        self.alignment_matrix = alignment_matrix      # Array2<i32>
        self.direction_matrix = direction_matrix      # Array2<Direction>
        self.optimal_alignment = optimal_alignment    # (Vec<Protein>, Vec<Protein>)
        self.max_f = max_f

    def get_alignment_matrix(self):
        return self.alignment_matrix

    def get_direction_matrix(self):
        return self.direction_matrix

    def get_optimal_alignment(self):
        return self.optimal_alignment

    def represent(self):
        """aligner_core.rs:20-26 / :51-57."""
        kind = "Local" if self.max_f is not None else "Global"
        return "Optimal %s Alignment:\n%s\n%s" % (kind, Protein.vec_to_str(self.optimal_alignment[0]),
                                                  Protein.vec_to_str(self.optimal_alignment[1]))


class GlobalAlignmentResult(_LegacyResult):
    pass


class LocalAlignmentResult(_LegacyResult):
    pass


class SimpleAligner:
    """aligner_core.rs:72-77.  from_seqs takes the raw FASTA record bytes (`seqs[i].seq`, test_alignment.rs:80-81); the
    u8 -> Protein converter lives in a module missing from the reference tree -- bytes outside the alphabet are dropped."""

    def __init__(self, seq_1, seq_2):
        self.u8_sequence_1 = bytes(seq_1)
        self.u8_sequence_2 = bytes(seq_2)
        self.protein_sequence_1 = self._to_protein(self.u8_sequence_1)
        self.protein_sequence_2 = self._to_protein(self.u8_sequence_2)

    @staticmethod
    def _to_protein(raw):
        enc, _ = Protein._tables()
        codes = enc[np.frombuffer(raw, dtype=np.uint8)]
        return codes[codes < Protein.volume()].copy()

    @classmethod
    def from_seqs(cls, seq_1, seq_2):
        return cls(seq_1, seq_2)

    def _run(self, semantics, del_, matrix, want_matrices):
        res, qa, ta, D, H = runtime.align_pair(semantics, self.protein_sequence_1, self.protein_sequence_2,
                                               float(int(del_)), float(int(del_)), matrix,
                                               want_directions=want_matrices, want_h=want_matrices)
        Hi = H.astype(np.int32) if H is not None else None
        return res, qa, ta, D, Hi

    def global_alignment(self, del_, matrix, want_matrices=True):
        """aligner_core.rs:96-183."""
        res, qa, ta, D, H = self._run(_ffi.LEGACY_GLOBAL, del_, matrix, want_matrices)
        return GlobalAlignmentResult(H, D, (qa, ta))

    def local_alignment(self, del_, matrix, want_matrices=True):
        """aligner_core.rs:185-269."""
        res, qa, ta, D, H = self._run(_ffi.LEGACY_LOCAL, del_, matrix, want_matrices)
        return LocalAlignmentResult(H, D, (qa, ta), max_f=int(res.score))

    def get_symbolic(self):
        return self.u8_sequence_1.decode("utf-8"), self.u8_sequence_2.decode("utf-8")
