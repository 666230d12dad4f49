"""SimpleGlobalAligner / SimpleLocalAligner -- the reference's `AlignerTrait` for the DP path, on the GPU.

Mirror of aligner-core/src/simple/mod.rs:9-40 (types + constructors), :42-145 (global perform_alignment) and
:147-264 (local), and of the trait in aligner-core/src/lib.rs:27-40.  Same names, argument meaning and error
behaviour; the matrix fill and the traceback run in the HIP kernels behind aln_align_pair.  `T` (the residue
type parameter) is the `alphabet` argument: aligner_amd.enums.Protein or .DNA.
"""
import numpy as np

from . import _ffi
from . import runtime
from .alignment import Alignment, AlignmentResult
from .enums import Protein
from .errors import AlignerError, ErrorKind


class Heuristics:
    """lib.rs:21-25.  Passing one to the simple aligners is an error (UnnecessaryArgument), as in the reference."""

    def __init__(self, kd, r_squared, frequencies):
        self.kd, self.r_squared, self.frequencies = kd, r_squared, frequencies


class _SimpleAligner:
    _semantics = None

    def __init__(self, query, target, alphabet=Protein):
        self.alphabet = alphabet
        self.query = np.array(query, dtype=np.uint8, copy=True)     # Vec::from(query) -- owns a copy
        self.target = np.array(target, dtype=np.uint8, copy=True)

    @classmethod
    def from_str_seqs(cls, query, target, alphabet=Protein):
        """simple/mod.rs:22-33 / :148-159 -- Err(CharIsNotMatchable) on any unknown character."""
        return cls(alphabet.str_to_vec(query), alphabet.str_to_vec(target), alphabet)

    @classmethod
    def from_seqs(cls, query, target, alphabet=Protein):
        """simple/mod.rs:35-40 / :161-166."""
        return cls(query, target, alphabet)

    def perform_alignment(self, del_, ext, matrix, heuristics=None, *, want_matrices=False, device=None, **kw):
        """perform_alignment(del, ext, &matrix, heuristics) -> Result<AlignmentResult<T, Alignment<T>>>.

        want_matrices=True also returns AlignmentResult.alignment_matrix / .direction_matrix (the reference
        always allocates them; nothing reads them, so they are opt-in here)."""
        res, qa, ta, D, H = runtime.align_pair(
            self._semantics, self.query, self.target, del_, ext, matrix, heuristics_present=heuristics is not None,
            want_directions=want_matrices, want_h=want_matrices, device=device, blank=self.alphabet.blank(), **kw)
        N, M = len(self.query), len(self.target)
        if self._semantics == _ffi.CORE_GLOBAL:
            coords = ((1, N), (1, M))                                # simple/mod.rs:138
        else:
            coords = ((res.start_x + 1, res.end_x + 1), (res.start_y + 1, res.end_y + 1))   # :255-258
        aln = Alignment(self.alphabet, qa, ta, coords, res.f)
        summary = {k: getattr(res, k) for k, _ in res._fields_}
        return AlignmentResult(aln, alignment_matrix=H, direction_matrix=D, matrix=None, score=res.score,
                               summary=summary)


class SimpleGlobalAligner(_SimpleAligner):
    """simple/mod.rs:9-12, impl :19-145."""
    _semantics = _ffi.CORE_GLOBAL


class SimpleLocalAligner(_SimpleAligner):
    """simple/mod.rs:14-17, impl :147-265."""
    _semantics = _ffi.CORE_LOCAL
