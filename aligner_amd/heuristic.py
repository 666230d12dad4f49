"""HeuristicAligner / HeuristicPWMAligner -- the matrix re-estimation loop around the DP path.

Mirror of aligner-core/src/heuristic/mod.rs:9-78 (pairwise) and :80-140 (PWM), with the helper numerics they call:
`transform_matrix` / `get_threshold` (aligner-helpers/src/matrices/mod.rs:8-68).  Each iteration is one hot-path call
with a REAL-VALUED matrix, so it runs on the f64 kernels (|max - x| < f64::EPSILON tie test, no fma contraction).
The loop itself is sequential (each matrix depends on the previous alignment) and stays on the host, as in the reference.

Third-party arithmetic not in the reference tree: roots 0.0.7 `find_roots_quadratic` (Cargo.lock:1781-1782), restated
below from its published algorithm (discriminant test, then the non-cancelling pair of formulas, roots in ascending
order).  No reference test covers this path: parity unpinned beyond source reading.
"""
import numpy as np

from .errors import AlignerError, ErrorKind, ReferencePanic
from .pwm import PWMAligner
from .simple import Heuristics, SimpleLocalAligner


class WrongMatrixSpecified(Exception):
    """`Err(aligner_helpers::Error::WrongMatrixSpecified)` (aligner-helpers/src/lib.rs:11-14)."""


def get_threshold(dim_1):
    """aligner-helpers/src/matrices/mod.rs:8-17."""
    return {20: 22.6, 21: 23.1, 22: 23.6, 23: 24.1, 24: 24.6}.get(dim_1, 0.0)


def find_roots_quadratic(a2, a1, a0):
    """roots 0.0.7: returns () / (x,) / (x_lo, x_hi)."""
    if a2 == 0.0:
        return () if a1 == 0.0 else (-a0 / a1,)
    disc = a1 * a1 - 4.0 * a2 * a0
    if disc < 0.0:
        return ()
    a2x2 = 2.0 * a2
    if disc == 0.0:
        return (-a1 / a2x2,)
    sq = np.sqrt(disc)
    same_sign, diff_sign = (-a1 + sq, -a1 - sq) if a1 < 0.0 else (-a1 - sq, -a1 + sq)
    if abs(same_sign) > abs(a2x2):
        a0x2 = 2.0 * a0
        if abs(diff_sign) > abs(a2x2):
            x1, x2 = a0x2 / same_sign, a0x2 / diff_sign
        else:
            x1, x2 = a0x2 / same_sign, same_sign / a2x2
    else:
        x1, x2 = diff_sign / a2x2, same_sign / a2x2
    return (x1, x2) if x1 < x2 else (x2, x1)


def transform_matrix(matrix, k_d, r_squared, frequencies):
    """aligner-helpers/src/matrices/mod.rs:19-68: rescale `matrix` so that its expectation under p = freq x uniform is k_d
    and its squared norm is r_squared."""
    m = np.asarray(matrix, dtype=np.float64)
    ncols = m.shape[1]
    f = np.full(ncols, 1.0 / ncols)
    p = np.outer(np.asarray(frequencies, dtype=np.float64), f)
    p_squared = (p * p).sum()
    k_0 = (p * m).sum()
    a = (k_d - k_0) / p_squared
    b = k_d / p_squared
    base = m + p * (a - b)
    denominator = (base * base).sum()
    a_coeff = (2.0 * b * (p * base).sum()) / denominator
    b_coeff = (b * b * p_squared - r_squared) / denominator
    roots = find_roots_quadratic(1.0, a_coeff, b_coeff)
    if len(roots) == 0:
        raise WrongMatrixSpecified()
    if len(roots) == 1:
        return p * b + roots[0] * base
    if roots[0] > 0.0 and roots[1] < 0.0:
        return p * b + roots[0] * base
    if roots[0] < 0.0 and roots[1] > 0.0:
        return p * b + roots[1] * base
    m1 = p * b + roots[0] * base
    m2 = p * b + roots[1] * base
    d1 = np.sqrt(((m - m1) ** 2).sum())
    d2 = np.sqrt(((m - m2) ** 2).sum())
    return m1 if d1 < d2 else m2


def _transform_or_panic(matrix, params):
    try:                                                                        # `.unwrap()` at heuristic/mod.rs:53, :71
        return transform_matrix(matrix, params.kd, params.r_squared, params.frequencies)
    except WrongMatrixSpecified:
        raise ReferencePanic(-1, "called `Result::unwrap()` on an `Err` value: WrongMatrixSpecified") from None


class _HeuristicLoop:
    def _loop(self, make_aligner, del_, ext, matrix, params, device):
        transformed = _transform_or_panic(matrix, params)
        max_f = 0.0
        while True:
            result = make_aligner().perform_alignment(del_, ext, transformed, None, device=device)
            if result.alignment.f > max_f:                                      # heuristic/mod.rs:64-72
                max_f = result.alignment.f
                transformed = _transform_or_panic(result.alignment.get_frequency_matrix(), params)
            else:
                result.matrix = transformed                                     # :73-75
                return result


class HeuristicAligner(_HeuristicLoop):
    """heuristic/mod.rs:9-78."""

    def __init__(self, query, target, alphabet):
        self.alphabet, self.query, self.target = alphabet, np.array(query, np.uint8), np.array(target, np.uint8)

    @classmethod
    def from_str_seqs(cls, query, target, alphabet):
        return cls(alphabet.str_to_vec(query), alphabet.str_to_vec(target), alphabet)

    @classmethod
    def from_seqs(cls, query, target, alphabet):
        return cls(query, target, alphabet)

    def perform_alignment(self, del_, ext, matrix, heuristics=None, device=None):
        if heuristics is None:
            raise AlignerError(ErrorKind.MissingArgument)                       # :42-45
        m = np.asarray(matrix, dtype=np.float64)
        params = Heuristics(heuristics.kd, heuristics.r_squared, heuristics.frequencies)
        if abs(params.r_squared - 0.0) < np.finfo(np.float64).eps:              # :47-49
            params.r_squared = float(m.shape[0] * m.shape[1])
        return self._loop(lambda: SimpleLocalAligner.from_seqs(self.query, self.target, self.alphabet), del_, ext, m,
                          params, device)


class HeuristicPWMAligner(_HeuristicLoop):
    """heuristic/mod.rs:80-140."""

    def __init__(self, query, alphabet):
        self.alphabet, self.query = alphabet, np.array(query, np.uint8)

    @classmethod
    def from_str_seqs(cls, query, _target, alphabet):
        return cls(alphabet.str_to_vec(query), alphabet)

    @classmethod
    def from_seqs(cls, query, _target, alphabet):
        return cls(query, alphabet)

    def perform_alignment(self, del_, ext, matrix, heuristics=None, device=None):
        if heuristics is None:
            raise AlignerError(ErrorKind.MissingArgument)
        return self._loop(lambda: PWMAligner.from_seqs(self.query, None, self.alphabet), del_, ext,
                          np.asarray(matrix, dtype=np.float64), heuristics, device)
