"""Batch driver behind calculate_p_value -- the reference's only in-tree batch site of the DP path.

Mirror of aligner-core/src/statistics/mod.rs:240-320: the reference spawns THREADS = 10 std::threads, each aligning
SEQUENCES / THREADS shuffled copies of the target against the query with SimpleLocalAligner and keeping only
`alignment.f` (:255-286; thread 5 does one alignment fewer so that, with the initial score, there are exactly 5000
samples).  Here the 4 999 shuffled targets are packed once and scored by ONE score-only batch call on the GPU
(no directions stored, no traceback).  The shuffling stays on the host (`thread_rng` in the reference is unseeded, so
parity for this function is statistical; parity of the scores for given shuffles is bit-exact and tested).

The extreme-value fit that turns the scores into a p-value (statistics/mod.rs:36-238) is host numerics in the
reference as well; it is restated in numpy below (including the loop-scoped re-binding of k / lambda at :69) and pinned
by tests/test_host_logic.py against an independent scalar restatement (tests/pyref.py).
"""
import numpy as np

from . import _ffi
from .batch import PairBatch, align_batch
from .errors import AlignerError, ErrorKind

MAXITER = 10000            # statistics/mod.rs:9-13
THREADS = 10
SEQUENCES = 5000
THRESHOLD_GLOBAL = 1e-6
THRESHOLD_LOCAL = 1e-4


def shuffle_and_randomize_sequence(sequence, rng):
    """statistics/mod.rs:309-320: drop 0..6 tail residues, then shuffle."""
    lock = int(rng.integers(0, 7))
    seq = np.array(sequence[:len(sequence) - lock], dtype=np.uint8, copy=True)
    rng.shuffle(seq)
    return seq


def shuffled_scores(query, target, initial_score, del_, ins, matrix, rng=None, device=None):
    """The scores / lengths vectors calculate_p_value builds (statistics/mod.rs:249-292), scored in one GPU batch.

    Returns (scores f64[5000], lengths int[5000], PairBatch of the 4 999 shuffled pairs)."""
    rng = rng or np.random.default_rng()
    query = np.asarray(query, dtype=np.uint8)
    target = np.asarray(target, dtype=np.uint8)
    pairs = []
    for i in range(THREADS):
        limit = SEQUENCES // THREADS
        if i == 5:
            limit = SEQUENCES - (SEQUENCES // THREADS * (THREADS - 1)) - 1      # :264-266
        for _ in range(limit):
            pairs.append((query, shuffle_and_randomize_sequence(target, rng)))
    batch = PairBatch.from_pairs(pairs)
    got = align_batch(batch, _ffi.CORE_LOCAL, del_, ins, matrix, device=device, want_traceback=False)
    bad = got.results["status"] != 0
    if bad.any():
        # the reference unwraps every perform_alignment (:273-277): a panic there is a panic here
        from .runtime import raise_for_status
        raise_for_status(int(got.results["status"][bad][0]), "calculate_p_value")
    scores = np.concatenate([[float(initial_score)], got.results["f"].astype(np.float64)])
    lengths = np.concatenate([[len(target)], batch.t_len.astype(np.int64)])
    return scores, lengths, batch


class DistributionParams:
    """statistics/mod.rs:15-34."""

    def __init__(self, k, lambda_, h):
        self.k, self.lambda_, self.h = k, lambda_, h

    def get_p_value(self, query_length, target_length, score):
        l = np.log(self.k * query_length * target_length) / self.h
        nn = (query_length - l) * (target_length - l)
        return 1.0 - np.exp(-self.k * nn * np.exp(-self.lambda_ * score))


def _nn(query_length, t, k, h):
    l = np.log(k * query_length * t) / h
    return (query_length - l) * (t - l)


def _estimate_k_and_lambda(query_length, t, scores, k, lam, h):
    """statistics/mod.rs:125-188 (Newton iteration on lambda, k from the normalisation)."""
    n = float(len(t))
    nn = _nn(query_length, t, k, h)
    es = np.exp(-lam * scores)
    s = (nn * es).sum()
    ws = (nn * scores * es).sum()
    with np.errstate(all="ignore"):
        for _ in range(MAXITER + 1):
            f = 1.0 / lam - scores.sum() / n + ws / s
            fd = -lam ** -2 - (nn * scores * scores * es).sum() / s + (ws / s) ** 2
            if not np.isfinite(f) or not np.isfinite(fd):
                return k, lam
            new_lam = lam - f / fd
            es = np.exp(-lam * scores)
            s = (nn * es).sum()
            ws = (nn * scores * es).sum()
            new_k = n / s
            if not np.isfinite(new_k) or new_k <= 0:
                return k, lam
            k, lam = new_k, new_lam
            if abs(f) < THRESHOLD_LOCAL:
                return k, lam
            nn = _nn(query_length, t, k, h)
    return k, lam


def _estimate_h(query_length, t, scores, k, lam, h):
    """statistics/mod.rs:190-238."""
    with np.errstate(all="ignore"):
        for _ in range(MAXITER + 1):
            l = np.log(k * query_length * t) / h
            nn = (query_length - l) * (t - l)
            a = 2.0 * l - query_length - t
            b = 1.0 / nn - k * np.exp(-lam * scores)
            c = -l / h
            g = (a * b * c).sum()
            gd = (2.0 * b * c * c - (a * c / nn) ** 2 - 2.0 * a * b * c / h).sum()
            if abs(g) < THRESHOLD_LOCAL:
                return h
            if gd > 0:
                h = h * 2.0 if g > 0 else h / 2.0
            elif g <= 0:
                h /= 2.0
            else:
                h -= g / gd
    return h


def calculate_distribution_params(query_length, target_lengths, scores):
    """statistics/mod.rs:36-123."""
    t_all = np.asarray(target_lengths, dtype=np.float64)
    s_all = np.asarray(scores, dtype=np.float64)
    if len(t_all) != len(s_all):
        raise AlignerError(ErrorKind.ValidationError)
    if len(s_all) == 0:
        raise AlignerError(ErrorKind.ValidationError)
    sd = ((s_all - s_all.mean()) ** 2).mean()          # central_moment(2), named `sd` in the reference
    lam = 1.0 / sd
    h = 1.0
    n = float(len(t_all))
    with np.errstate(all="ignore"):
        nn = query_length * t_all
        k = n / (nn * np.exp(-lam * s_all)).sum()
        ll = n * np.log(lam * k) + (np.log(nn) - lam * s_all - k * nn * np.exp(-lam * s_all)).sum()
        t_act, s_act = t_all.copy(), s_all.copy()
        # statistics/mod.rs:69: `let (k, lambda) = estimate_..(.., k, lambda, h)` SHADOWS inside the loop body, so every
        # iteration restarts the Newton step from the initial moment estimates k0 / lam0 (only h and the active subset
        # carry over), and the fall-through Ok(..) after MAXITER (:122) returns k0 / lam0 with the last h.
        k0, lam0 = k, lam
        for _ in range(MAXITER + 1):
            k, lam = _estimate_k_and_lambda(query_length, t_act, s_act, k0, lam0, h)
            h = _estimate_h(query_length, t_act, s_act, k, lam, h)
            nn = _nn(query_length, t_all, k, h)
            ll_new = n * np.log10(lam * k) + (np.log10(nn) - lam * s_all - k * nn * np.exp(-lam * s_all)).sum()
            if abs(ll_new - ll) / ll < THRESHOLD_GLOBAL:
                return DistributionParams(k, lam, h)
            ll = ll_new
            keep = n * (1.0 - np.exp(-k * nn * np.exp(-lam * s_all))) >= 1.0
            t_act, s_act = t_all[keep], s_all[keep]
    return DistributionParams(k0, lam0, h)


def calculate_p_value(query, target, initial_score, del_, ins, matrix, rng=None, device=None):
    """statistics/mod.rs:240-307."""
    scores, lengths, _ = shuffled_scores(query, target, initial_score, del_, ins, matrix, rng=rng, device=device)
    params = calculate_distribution_params(len(query), lengths, scores)
    return float(params.get_p_value(len(query), len(target), initial_score))
