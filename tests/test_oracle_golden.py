"""Pins the CPU oracle: against the reference's own golden matrices (src/tests/test_alignment.rs via
tests/golden/legacy_kat.json), against an independent pure-Python restatement on small seeded cases, and against
the restatement-derived anchors of SURVEY.md Appendix B.  No GPU."""
import numpy as np
import pytest

import pyref
from aligner_amd.enums import Protein

P = Protein.str_to_vec


def _H(d, M, N):
    return np.array([[d[(y, x)] for x in range(N + 1)] for y in range(M + 1)], dtype=np.float64)


def test_golden_legacy_global(orc, kat):
    """src/tests/test_alignment.rs:9-99."""
    q, t = P(kat["query"]), P(kat["target"])
    r = orc.align(orc.LEGACY_GLOBAL, q, t, kat["gap"], kat["gap"], kat["matrix"], want_matrices=True)
    assert r["status"] == 0
    assert (r["H"] == np.array(kat["global"]["H"])).all()
    assert (r["D"] == np.array(kat["global"]["D"])).all()
    assert r["qa"].tolist() == kat["global"]["query_aligned"]
    assert r["ta"].tolist() == kat["global"]["target_aligned"]


def test_golden_legacy_local(orc, kat):
    """src/tests/test_alignment.rs:101-191."""
    q, t = P(kat["query"]), P(kat["target"])
    r = orc.align(orc.LEGACY_LOCAL, q, t, kat["gap"], kat["gap"], kat["matrix"], want_matrices=True)
    assert (r["H"] == np.array(kat["local"]["H"])).all()
    assert (r["D"] == np.array(kat["local"]["D"])).all()
    assert r["qa"].tolist() == kat["local"]["query_aligned"]
    assert r["ta"].tolist() == kat["local"]["target_aligned"]
    assert r["score"] == 28 and r["end"] == (5, 9)


def test_golden_pins_core_global_when_del_equals_ext(orc, kat):
    """Core global with del == ext == 8 is the same recurrence as the legacy global test: H and D must equal the
    golden matrices; only the traceback differs (it starts AT (M,N) and so duplicates the seed pair)."""
    q, t = P(kat["query"]), P(kat["target"])
    r = orc.align(orc.CORE_GLOBAL, q, t, 8, 8, kat["matrix"], want_matrices=True)
    assert (r["H"] == np.array(kat["global"]["H"], dtype=np.float64)).all()
    assert (r["D"] == np.array(kat["global"]["D"])).all()
    assert Protein.vec_to_str(r["qa"]) == "HEAGAWGHE_EE"
    assert Protein.vec_to_str(r["ta"]) == "_PA__W_HEAEE"
    assert r["f"] == 0.0 and r["score"] == 1.0 and r["coords"] == ((1, 10), (1, 7))


def test_appendix_b_anchors(orc, blosum62, kat):
    q, t = P("HEAGAWGHEE"), P("PAWHEAE")
    r = orc.align(orc.CORE_LOCAL, q, t, 11, 2, blosum62)
    assert (r["f"], r["coords"]) == (27.0, ((5, 11), (1, 8)))
    assert (Protein.vec_to_str(r["qa"]), Protein.vec_to_str(r["ta"])) == ("_AWGHE_EE", "PAW_HEAEE")
    assert Protein.vec_to_str(orc.midline(r["qa"], r["ta"], blosum62)) == "_AW_HE_EE"
    r = orc.align(orc.CORE_GLOBAL, q, t, 11, 2, blosum62)
    assert (r["score"], r["f"], r["coords"]) == (21.0, 0.0, ((1, 10), (1, 7)))
    assert (Protein.vec_to_str(r["qa"]), Protein.vec_to_str(r["ta"])) == ("HEAGAWGHE_EE", "P_A__W_HEAEE")
    assert Protein.vec_to_str(orc.midline(r["qa"], r["ta"], blosum62)) == "__A__W_HE_EE"
    r = orc.align(orc.CORE_LOCAL, q, t, 8, 8, kat["matrix"], want_matrices=True)
    assert (r["f"], r["coords"]) == (26.0, ((4, 10), (1, 6)))
    assert (Protein.vec_to_str(r["qa"]), Protein.vec_to_str(r["ta"])) == ("GAWGHEE", "PAW_HEE")
    assert r["H"].min() == -7.0          # no clamp at zero in the core "local" (enums.rs:30-46)


@pytest.mark.parametrize("seed", range(12))
def test_oracle_matches_python_restatement(orc, blosum62, seed):
    rng = np.random.default_rng(seed)
    N, M = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    A = 4 if seed % 3 == 0 else 24
    q = rng.integers(0, A, N).astype(np.uint8)
    t = rng.integers(0, A, M).astype(np.uint8)
    if seed % 3 == 0:   # zero-rich adversarial scoring: +-1 matrix, small gaps (row-1 hazard territory)
        S = np.where(np.eye(24) > 0, 1.0, -1.0)
        dele, ext = [(2, 1), (3, 1), (1, 2), (1, 1)][seed % 4]
    else:
        S = blosum62
        dele, ext = [(11, 2), (11, 1), (8, 8)][seed % 3]
    Sl = S.tolist()
    for sem, local in ((orc.CORE_GLOBAL, False), (orc.CORE_LOCAL, True)):
        ref = pyref.core(q.tolist(), t.tolist(), float(dele), float(ext), Sl, local)
        got = orc.align(sem, q, t, dele, ext, S, want_matrices=True)
        if ref.get("panic"):
            assert got["status"] == orc.ERR_NO_POSITIVE_CELL
            continue
        assert (got["H"] == _H(ref["H"], M, N)).all()
        assert (got["D"] == _H(ref["D"], M, N)).all()
        assert got["qa"].tolist() == ref["qa"] and got["ta"].tolist() == ref["ta"]
        assert got["f"] == ref["f"] and got["coords"] == ref["coords"] and got["score"] == ref["score"]
    for sem, local in ((orc.LEGACY_GLOBAL, False), (orc.LEGACY_LOCAL, True)):
        ref = pyref.legacy(q.tolist(), t.tolist(), int(dele), Sl, local)
        got = orc.align(sem, q, t, dele, dele, S, want_matrices=True)
        assert (got["H"] == _H(ref["H"], M, N)).all()
        assert (got["D"] == _H(ref["D"], M, N)).all()
        assert got["qa"].tolist() == ref["qa"] and got["ta"].tolist() == ref["ta"]
        assert got["score"] == ref["score"]


def test_oracle_f64_matrix(orc):
    """Real-valued matrices (the heuristic aligner's use, heuristic/mod.rs:52-62) go through the same f64 code."""
    rng = np.random.default_rng(5)
    S = rng.normal(0, 2, (24, 24))
    q = rng.integers(0, 24, 30).astype(np.uint8)
    t = rng.integers(0, 24, 25).astype(np.uint8)
    ref = pyref.core(q.tolist(), t.tolist(), 3.7, 0.9, S.tolist(), True)
    got = orc.align(orc.CORE_LOCAL, q, t, 3.7, 0.9, S, want_matrices=True)
    assert (got["H"] == _H(ref["H"], 25, 30)).all()
    assert got["qa"].tolist() == ref["qa"] and got["f"] == ref["f"]


def test_oracle_error_paths(orc, blosum62):
    q, t = P("HEAG"), P("PAW")
    assert orc.align(orc.CORE_LOCAL, q, t, 11, 2, blosum62, heuristics_present=True)["status"] == orc.ERR_UNNECESSARY_ARGUMENT
    assert orc.align(orc.CORE_GLOBAL, q[:0], t, 11, 2, blosum62)["status"] == orc.ERR_EMPTY_SEQUENCE
    assert orc.align(orc.CORE_GLOBAL, q, t, 11, 2, blosum62[:3, :3])["status"] == orc.ERR_CODE_OUT_OF_RANGE
    # all-mismatch: no positive cell -> the reference's argmax lands on (0,0) and it panics
    S = -np.ones((24, 24))
    assert orc.align(orc.CORE_LOCAL, q, t, 11, 2, S)["status"] == orc.ERR_NO_POSITIVE_CELL


def test_oracle_batch_threads_equal_single(orc, blosum62):
    from aligner_amd import workloads
    b = workloads.c5_batch(n_pairs=24, lo=20, hi=120)
    r1, tb1, off = orc.align_batch(orc.CORE_LOCAL, b.seqs, b.q_off, b.q_len, b.t_off, b.t_len, 11, 2, blosum62, 1)
    r4, tb4, _ = orc.align_batch(orc.CORE_LOCAL, b.seqs, b.q_off, b.q_len, b.t_off, b.t_len, 11, 2, blosum62, 4)
    assert (tb1 == tb4).all()
    for i in range(len(b)):
        one = orc.align(orc.CORE_LOCAL, b.query(i), b.target(i), 11, 2, blosum62)
        assert (r1[i].f, r1[i].aln_len, r1[i].status) == (one["f"], len(one["qa"]), one["status"])
        assert (r4[i].f, r4[i].aln_len) == (r1[i].f, r1[i].aln_len)


@pytest.mark.parametrize("seed", range(6))
def test_oracle_pwm_matches_python_restatement(orc, seed):
    """PWMAligner (pwm/mod.rs:29-126) has no reference test: parity unpinned beyond source reading; the C oracle and an
    independent Python restatement must at least agree."""
    rng = np.random.default_rng(100 + seed)
    Q, W = int(rng.integers(1, 45)), int(rng.integers(1, 40))
    seq = rng.integers(0, 4, Q).astype(np.uint8)
    M = rng.integers(-1, 2, (4, W)).astype(np.float64) if seed % 2 == 0 else np.round(rng.normal(0, 1, (4, W)), 2)
    dele, ext = [(3, 1), (1, 2), (2, 2)][seed % 3]
    ref = pyref.pwm(seq.tolist(), float(dele), float(ext), M.tolist())
    got = orc.align_pwm(seq, dele, ext, M, want_matrices=True)
    assert (got["H"] == _H(ref["H"], Q, W)).all() and (got["D"] == _H(ref["D"], Q, W)).all()
    assert got["numbered"].tolist() == ref["numbered"] and got["qal"].tolist() == ref["qal"]
    assert got["f"] == ref["f"] and got["coords"] == ref["coords"]
    assert orc.align_pwm(seq, dele, ext, M[:3])["status"] == orc.ERR_MATRIX_SHAPE
    assert orc.align_pwm(seq, dele, ext, M, heuristics_present=True)["status"] == orc.ERR_UNNECESSARY_ARGUMENT
