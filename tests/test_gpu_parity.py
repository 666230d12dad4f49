"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.
Bit-exact: integer scores, coordinates, every direction, both aligned strings.  Needs an MI355X."""
import numpy as np
import pytest

from aligner_amd import _ffi, runtime, workloads
from aligner_amd.batch import PairBatch, align_batch
from aligner_amd.enums import DNA, Protein
from aligner_amd.errors import AlignerError, ErrorKind, ReferencePanic
from aligner_amd.legacy import SimpleAligner
from aligner_amd.matrices import nucleotide_matrix
from aligner_amd.simple import Heuristics, SimpleGlobalAligner, SimpleLocalAligner

pytestmark = pytest.mark.gpu
P = Protein.str_to_vec
SEMS = [_ffi.CORE_GLOBAL, _ffi.CORE_LOCAL, _ffi.LEGACY_GLOBAL, _ffi.LEGACY_LOCAL]


def check_pair(orc, sem, q, t, dele, ext, S, full=True, directions_only=False, **kw):
    """GPU vs oracle on one pair: summary, strings and (optionally) the whole H and D matrices.
    directions_only: compare D but do not request H (the H dump is served by the generic kernels only)."""
    ref = orc.align(sem, q, t, dele, ext, S, want_matrices=full)
    if ref["status"] != 0:
        with pytest.raises(ReferencePanic) as e:
            runtime.align_pair(sem, q, t, dele, ext, S, **kw)
        assert e.value.status == ref["status"]
        return None
    res, qa, ta, D, H = runtime.align_pair(sem, q, t, dele, ext, S, want_directions=full,
                                           want_h=full and not directions_only, **kw)
    assert res.status == 0
    assert res.score == ref["score"], (res.score, ref["score"])
    assert res.f == ref["f"]
    assert (res.end_y, res.end_x) == ref["end"]
    assert (res.start_y, res.start_x) == ref["start"]
    assert qa.tolist() == ref["qa"].tolist()
    assert ta.tolist() == ref["ta"].tolist()
    if full:
        if H is not None:
            bad = np.argwhere(H != ref["H"])
            assert len(bad) == 0, "H differs first at (y,x)=%s: gpu %s ref %s" % (bad[0], H[tuple(bad[0])], ref["H"][tuple(bad[0])])
        bad = np.argwhere(D != ref["D"])
        assert len(bad) == 0, "D differs first at (y,x)=%s" % (bad[0],)
    return res


def test_device_is_mi355x():
    info = runtime.device_info()
    assert info["compute_units"] >= 64
    assert "gfx950" in info["name"], info


# ---------------------------------------------------------------- the reference's own known answers
def test_kat_legacy_global_on_gpu(kat):
    """src/tests/test_alignment.rs:9-99 through SimpleAligner.global_alignment."""
    r = SimpleAligner.from_seqs(b"HEAGAWGHEE", b"PAWHEAE").global_alignment(8, kat["matrix"])
    assert (r.get_alignment_matrix() == np.array(kat["global"]["H"])).all()
    assert (r.get_direction_matrix() == np.array(kat["global"]["D"])).all()
    assert r.get_optimal_alignment()[0].tolist() == kat["global"]["query_aligned"]
    assert r.get_optimal_alignment()[1].tolist() == kat["global"]["target_aligned"]


def test_kat_legacy_local_on_gpu(kat):
    """src/tests/test_alignment.rs:101-191 through SimpleAligner.local_alignment."""
    r = SimpleAligner.from_seqs(b"HEAGAWGHEE", b"PAWHEAE").local_alignment(8, kat["matrix"])
    assert (r.get_alignment_matrix() == np.array(kat["local"]["H"])).all()
    assert (r.get_direction_matrix() == np.array(kat["local"]["D"])).all()
    assert r.get_optimal_alignment()[0].tolist() == kat["local"]["query_aligned"]
    assert r.get_optimal_alignment()[1].tolist() == kat["local"]["target_aligned"]
    assert r.max_f == 28


def test_kat_core_global_del_eq_ext(kat):
    r = SimpleGlobalAligner.from_str_seqs("HEAGAWGHEE", "PAWHEAE").perform_alignment(8.0, 8.0, kat["matrix"],
                                                                                     want_matrices=True)
    assert (r.alignment_matrix == np.array(kat["global"]["H"], dtype=np.float64)).all()
    assert (r.direction_matrix == np.array(kat["global"]["D"])).all()
    assert r.alignment.query_str() == "HEAGAWGHE_EE" and r.alignment.target_str() == "_PA__W_HEAEE"
    assert r.alignment.f == 0.0 and r.alignment.coords == ((1, 10), (1, 7))


def test_cli_plumbing_book_example(blosum62):
    """C1: aligner-cli defaults (del 11 / ext 2, BLOSUM62) on examples/book_example_1.fasta, both modes."""
    g = SimpleGlobalAligner.from_str_seqs("HEAGAWGHEE", "PAWHEAE").perform_alignment(11.0, 2.0, blosum62)
    assert (g.alignment.query_str(), g.alignment.target_str()) == ("HEAGAWGHE_EE", "P_A__W_HEAEE")
    assert g.alignment.midline_str(blosum62) == "__A__W_HE_EE" and g.score == 21.0
    l = SimpleLocalAligner.from_str_seqs("HEAGAWGHEE", "PAWHEAE").perform_alignment(11.0, 2.0, blosum62)
    assert (l.alignment.query_str(), l.alignment.target_str()) == ("_AWGHE_EE", "PAW_HEAEE")
    assert l.alignment.f == 27.0 and l.alignment.coords == ((5, 11), (1, 8))


@pytest.mark.parametrize("sem", [_ffi.CORE_GLOBAL, _ffi.CORE_LOCAL, _ffi.LEGACY_LOCAL])
def test_c_harness_through_the_header(orc, blosum62, tmp_path, sem):
    """include/aligner_hip.h compiled as C (tests/abi_harness.c): the book example through aln_align_pair with every output, and
    (q, t), (t, q), (q, q) through aln_align_batch, printed by the C program and compared with the oracle here."""
    import subprocess
    from aligner_amd import build as native_build
    exe = native_build.build_harness()
    q, t = P("HEAGAWGHEE"), P("PAWHEAE")
    dele, ext = (8, 8) if sem == _ffi.LEGACY_LOCAL else (11, 2)
    case = tmp_path / "case.txt"
    case.write_text("%d %g %g 24 24\n%s\n%d %s\n%d %s\n" % (sem, dele, ext, " ".join("%g" % v for v in blosum62.ravel()),
                                                         len(q), " ".join(map(str, q.tolist())), len(t), " ".join(map(str, t.tolist()))))
    out = subprocess.run([exe, str(case)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    facts = {}
    for line in out.stdout.splitlines():
        k, _, v = line.partition(" ")
        facts[k] = v.split()

    def summary(words):
        d = dict(zip(words[0::2], words[1::2])) if words[4] != "end" else None
        return d
    def check(prefix, qq, tt, full):
        ref = orc.align(sem, qq, tt, dele, ext, blosum62, want_matrices=full)
        w = facts[prefix]
        assert w[0] == "status" and int(w[1]) == ref["status"]
        assert float(w[3]) == ref["f"] and float(w[5]) == ref["score"]
        assert (int(w[7]), int(w[8])) == ref["end"] and (int(w[10]), int(w[11])) == ref["start"]
        assert [int(v) for v in facts[prefix + "_q_aln"]] == ref["qa"].tolist()
        assert [int(v) for v in facts[prefix + "_t_aln"]] == ref["ta"].tolist()
        assert int(w[13]) == len(ref["qa"])
        if full:
            assert [int(v) for v in facts[prefix + "_dirs"]] == ref["D"].ravel().tolist()
            assert [float(v) for v in facts[prefix + "_h"]] == ref["H"].ravel().astype(float).tolist()
    check("pair", q, t, True)
    check("batch0", q, t, False)
    check("batch1", t, q, False)
    check("batch2", q, q, False)


# ---------------------------------------------------------------- differential tests vs the oracle
@pytest.mark.parametrize("sem", SEMS)
@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (9, 1), (10, 7), (64, 64), (65, 63), (130, 129), (257, 70),
                                   (200, 513), (90, 1030),
                                   # last strips of every height class: R = 3, 4, 5, 6, 7, 8 rows per lane (aln_pick_r)
                                   (110, 190), (75, 250), (100, 300), (120, 350), (90, 420), (70, 500), (150, 812), (80, 940)])
def test_random_protein_full_matrix(orc, blosum62, sem, shape):
    """Full H and D matrices from the generic integer kernels (the H dump routes there), summary + directions +
    strings from the fast (query-profile) integer kernels."""
    N, M = shape
    rng = np.random.default_rng(N * 1000 + M)
    q = rng.integers(0, 20, N).astype(np.uint8)
    t = rng.integers(0, 20, M).astype(np.uint8)
    check_pair(orc, sem, q, t, 11, 2, blosum62)
    check_pair(orc, sem, q, t, 11, 2, blosum62, directions_only=True)
    check_pair(orc, sem, q, t, 11, 2, blosum62, directions_only=True, force_generic=True)


@pytest.mark.parametrize("sem", [_ffi.CORE_GLOBAL, _ffi.CORE_LOCAL])
@pytest.mark.parametrize("shape", [(33, 40), (150, 150), (300, 600)])
def test_f64_kernels_match_oracle(orc, blosum62, sem, shape):
    """f64 kernels: forced on integral data, and on a real-valued matrix (the heuristic aligner's input)."""
    N, M = shape
    rng = np.random.default_rng(7 + N)
    q = rng.integers(0, 24, N).astype(np.uint8)
    t = rng.integers(0, 24, M).astype(np.uint8)
    res = check_pair(orc, sem, q, t, 11, 2, blosum62, force_f64=True)
    assert res is None or (res.flags & 1) == 0
    S = np.round(rng.normal(0, 2.5, (24, 24)), 3)
    S[np.arange(24), np.arange(24)] = np.abs(S[np.arange(24), np.arange(24)]) + 2.0
    res = check_pair(orc, sem, q, t, 3.7, 0.9, S)
    assert res is None or (res.flags & 1) == 0


@pytest.mark.parametrize("sem", [_ffi.CORE_GLOBAL, _ffi.CORE_LOCAL])
@pytest.mark.parametrize("gaps", [(11.3, 2.1), (3.3, 3.3)])
def test_real_valued_batch_on_the_lean_f64_strip(orc, blosum62, monkeypatch, sem, gaps):
    """Real-valued batches of the core semantics run aln_fill_f64_kernel (the lean f64 strip: asm cell blocks, tags through carry
    masks, wave-wide end-cell threshold).  90 pairs of 1..1400 residues (one to three strips, every R of the last
    strip), related and unrelated, BLOSUM62 x 0.37: summaries and strings against the oracle, and record for record -- the number
    of advice passes included -- against run_strip's loop in the same build (ALN_F64_OLD=1: a scalar condition code lost across an
    asm block once showed only as extra passes)."""
    rng = np.random.default_rng(370 + sem)
    pairs = []
    for i in range(90):
        N, M = int(rng.integers(1, 1400)), int(rng.integers(1, 1400))
        q = rng.integers(0, 20, N).astype(np.uint8)
        t = rng.integers(0, 20, M).astype(np.uint8)
        if i % 2 and min(N, M) > 8:
            L = min(N, M) // 2
            t[M // 4:M // 4 + L] = q[N // 4:N // 4 + L][:len(t[M // 4:M // 4 + L])]
        pairs.append((q, t))
    b = PairBatch.from_pairs(pairs)
    S = blosum62 * 0.37
    got = _check_batch(orc, b, sem, gaps[0], gaps[1], S)
    assert not (got.results["flags"] & 1).any()
    monkeypatch.setenv("ALN_F64_OLD", "1")
    old = align_batch(b, sem, gaps[0], gaps[1], S)
    for f in ("status", "score", "f", "end_y", "end_x", "start_y", "start_x", "aln_len", "passes"):
        assert (got.results[f] == old.results[f]).all(), f


@pytest.mark.parametrize("sem", SEMS)
@pytest.mark.parametrize("shape", [(300, 600), (1000, 1000), (260, 65), (700, 1500), (2100, 2048), (16, 1024)])
def test_generic_pair_filled_by_one_workgroup(orc, blosum62, sem, shape):
    """A single pair of the generic kernels (real-valued matrix: every HeuristicAligner iteration; or integers forced off the fast
    path) is filled by one workgroup, one wave per strip of 64 R rows, the strips pipelined through LDS rings (flags bit 2):
    summary, both strings and every direction against the oracle.  R = 2 above 512 rows, 4 above 1024; N > 256 wraps the rings."""
    N, M = shape
    rng = np.random.default_rng(N * 7 + M)
    q = rng.integers(0, 20, N).astype(np.uint8)
    t = rng.integers(0, 20, M).astype(np.uint8)
    L = min(N, M) // 2
    t[:L] = q[:L]                                                    # a long diagonal: the path crosses several strips
    S = np.round(blosum62 * 0.5 + rng.normal(0, 0.05, blosum62.shape), 3)
    legacy = sem in (_ffi.LEGACY_GLOBAL, _ffi.LEGACY_LOCAL)
    if not legacy:                                                   # the legacy aligner is integer-only (aligner_core.rs)
        res = check_pair(orc, sem, q, t, 11.5, 2.25, S, directions_only=True)
        assert res is None or ((res.flags & 4) and not (res.flags & 1))
    res = check_pair(orc, sem, q, t, 11, 11 if legacy else 2, blosum62, directions_only=True, force_generic=True)
    assert res is None or (res.flags & 4)
    if N * M <= 400000 and not legacy:
        res = check_pair(orc, sem, q, t, 11, 2, blosum62, full=False, force_f64=True)
        assert res is None or (res.flags & 4)


@pytest.mark.parametrize("gaps", [(2, 1), (1, 2), (3, 1)])
@pytest.mark.parametrize("shape", [(200, 200), (400, 130), (300, 700), (257, 66)])
def test_generic_pair_one_workgroup_row1_hazard(orc, gaps, shape):
    """The same route where the row-1 hazard bites (4 letters, +-1 scores, del != ext: exact zeros everywhere): the workgroup
    repeats the pass with the advice it observed until it is self-consistent, or ends in the strict-order routine."""
    N, M = shape
    rng = np.random.default_rng(N + 31 * M + gaps[0])
    q = rng.integers(0, 4, N).astype(np.uint8)
    t = rng.integers(0, 4, M).astype(np.uint8)
    S = np.where(np.eye(4) > 0, 1.0, -1.0)
    seen = 0
    for kw in (dict(force_generic=True), dict(force_f64=True)):
        res = check_pair(orc, _ffi.CORE_LOCAL, q, t, gaps[0], gaps[1], S, directions_only=True, **kw)
        if res is not None:
            assert res.flags & 4
            seen = max(seen, res.passes & 0xff)
    assert seen >= 1
    # one pass allowed: the advice is not self-consistent, the workgroup's first thread runs the strict-order routine (row-major
    # directions, walked by the batch traceback kernels)
    res = check_pair(orc, _ffi.CORE_LOCAL, q, t, gaps[0], gaps[1], S, directions_only=True, force_f64=True, max_passes=1)
    if res is not None and seen >= 2:
        assert (res.flags & 4) and (res.passes & 0x80)


def test_generic_kernels_accept_a_pass_whose_new_advice_moves_no_cell_of_row_1(orc, blosum62):
    """Real-valued scores: the bottom row has exact zeros at arbitrary columns, so the all-"ext" advice of the first pass is
    wrong somewhere in nearly every pair -- but only row 1 reads it.  When the corrected advice leaves every cell of row 1
    as the pass computed it, the pass is the answer (adopt_advice_checked) and is not repeated: single pairs (one workgroup
    per pair) and a batch (one wave per pair) against the oracle, and both outcomes -- one pass, two passes -- must occur."""
    rng = np.random.default_rng(2718)
    S = blosum62 * 0.5
    one = two = 0
    pairs = []
    for i in range(10):
        N, M = int(rng.integers(300, 900)), int(rng.integers(300, 900))
        q = rng.integers(0, 20, N).astype(np.uint8)
        t = rng.integers(0, 20, M).astype(np.uint8)
        pairs.append((q, t))
        ref = orc.align(_ffi.CORE_LOCAL, q, t, 11.5, 2.25, S, want_matrices=True)
        zeros = bool((ref["H"][M, 1:N] == 0).any())                  # the first pass's advice is not the final one
        res = check_pair(orc, _ffi.CORE_LOCAL, q, t, 11.5, 2.25, S, directions_only=True, force_f64=True)   # (a dyadic scheme: the integer kernels would take it)
        assert res.flags & 4
        if zeros and (res.passes & 0x7f) == 1:
            one += 1
        if (res.passes & 0x7f) >= 2:
            two += 1
    assert one >= 1 and two >= 1, (one, two)
    b = PairBatch.from_pairs(pairs * 3)                              # 30 pairs: the batch kernel
    got = _check_batch(orc, b, _ffi.CORE_LOCAL, 11.5, 2.25, S, force_f64=True)
    p = got.results["passes"] & 0x7f
    assert (p == 1).any() and (p >= 2).any()


@pytest.mark.parametrize("sem", SEMS)
def test_serial_order_kernel(orc, blosum62, sem):
    rng = np.random.default_rng(11)
    q = rng.integers(0, 20, 77).astype(np.uint8)
    t = rng.integers(0, 20, 140).astype(np.uint8)
    res = check_pair(orc, sem, q, t, 11, 2, blosum62, force_serial=True)
    assert res.passes & 0x80


@pytest.mark.parametrize("gaps", [(2, 1), (3, 1), (1, 2), (11, 2)])
@pytest.mark.parametrize("shape", [(200, 200), (200, 60), (300, 3), (257, 2), (64, 1), (500, 700)])
def test_row1_hazard_zero_rich(orc, gaps, shape):
    """CORE_LOCAL with del != ext: the penalty of cell (1,x) depends on the bottom cell of column x-1.
    4-letter alphabet, +-1 scoring makes exact zeros (and bottom-row zeros) frequent; tiny M needs several passes."""
    N, M = shape
    rng = np.random.default_rng(N + 31 * M + gaps[0])
    q = rng.integers(0, 4, N).astype(np.uint8)
    t = rng.integers(0, 4, M).astype(np.uint8)
    S = np.where(np.eye(4) > 0, 1.0, -1.0)
    check_pair(orc, _ffi.CORE_LOCAL, q, t, gaps[0], gaps[1], S)
    check_pair(orc, _ffi.CORE_LOCAL, q, t, gaps[0], gaps[1], S, directions_only=True)
    check_pair(orc, _ffi.CORE_LOCAL, q, t, gaps[0], gaps[1], S, max_passes=1)   # forces the serial fallback when needed
    check_pair(orc, _ffi.CORE_LOCAL, q, t, gaps[0], gaps[1], S, directions_only=True, max_passes=1)


def test_homopolymer_and_ties(orc, blosum62):
    for q, t in ((np.zeros(300, np.uint8), np.zeros(280, np.uint8)),
                 (np.tile(np.array([0, 1, 2, 3], np.uint8), 90), np.tile(np.array([0, 1, 2, 3], np.uint8), 70))):
        for sem in SEMS:
            check_pair(orc, sem, q, t, 11, 2, blosum62)
            check_pair(orc, sem, q, t, 4, 4, blosum62)
            check_pair(orc, sem, q, t, 11, 2, blosum62, directions_only=True)
            check_pair(orc, sem, q, t, 4, 4, blosum62, directions_only=True)


def test_c2_config_1k_pair(orc, blosum62):
    """BASELINE C2: 1k x 1k protein, core local, 11/2 and 11/1, random and homolog variants; legacy local gap 11."""
    for homolog in (False, True):
        q, t = workloads.c2_pair(homolog)
        for ext in (2, 1):
            check_pair(orc, _ffi.CORE_LOCAL, q, t, 11, ext, blosum62)
            check_pair(orc, _ffi.CORE_LOCAL, q, t, 11, ext, blosum62, directions_only=True)
        check_pair(orc, _ffi.LEGACY_LOCAL, q, t, 11, 11, blosum62)
        check_pair(orc, _ffi.LEGACY_LOCAL, q, t, 11, 11, blosum62, directions_only=True)
        check_pair(orc, _ffi.LEGACY_GLOBAL, q, t, 11, 11, blosum62, directions_only=True)
        check_pair(orc, _ffi.CORE_GLOBAL, q, t, 11, 2, blosum62, directions_only=True)
        check_pair(orc, _ffi.CORE_GLOBAL, q, t, 11, 2, blosum62, full=False)


# ---------------------------------------------------------------- batch driver
def _check_batch(orc, b, sem, dele, ext, S, **kw):
    got = align_batch(b, sem, dele, ext, S, **kw)
    ref, tb, tb_off = orc.align_batch(sem, b.seqs, b.q_off, b.q_len, b.t_off, b.t_len, dele, ext, S, n_threads=8)
    for i in range(len(b)):
        r, g = ref[i], got.results[i]
        assert g["status"] == r.status, i
        if r.status != 0:
            continue
        assert (g["score"], g["f"], g["end_y"], g["end_x"], g["start_y"], g["start_x"], g["aln_len"]) == \
               (r.score, r.f, r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len), i
        cap = int(b.q_len[i] + b.t_len[i]) + 2
        o = int(tb_off[i])
        qa, ta = got.aligned(i)
        assert (qa == tb[o:o + r.aln_len]).all() and (ta == tb[o + cap:o + cap + r.aln_len]).all(), i
    return got


def test_batch_mixed_protein(orc, blosum62):
    b = workloads.c5_batch(n_pairs=300, lo=20, hi=700)
    got = _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    assert (got.results["flags"] & 1).all()
    _check_batch(orc, b, _ffi.CORE_GLOBAL, 11, 2, blosum62)


def test_batch_nucleotide_c3_shape(orc):
    """BASELINE C3 shape at reduced count: 150 bp read pairs, core global, +5/-4, del 10 / ext 1."""
    b = workloads.c3_batch(n_pairs=500)
    _check_batch(orc, b, _ffi.CORE_GLOBAL, 10, 1, nucleotide_matrix())
    _check_batch(orc, b, _ffi.CORE_LOCAL, 10, 1, nucleotide_matrix())


def test_batch_with_invalid_pairs(orc, blosum62):
    """Empty sequences and out-of-range codes are per-pair statuses; the rest of the batch is unaffected."""
    rng = np.random.default_rng(3)
    pairs = [(rng.integers(0, 20, 50).astype(np.uint8), rng.integers(0, 20, 60).astype(np.uint8)) for _ in range(6)]
    pairs[1] = (np.zeros(0, np.uint8), pairs[1][1])
    pairs[3] = (pairs[3][0], np.array([1, 2, 77, 3], np.uint8))
    pairs[4] = (np.array([5], np.uint8), np.array([4], np.uint8))     # Q-C scores -3: no positive cell
    b = PairBatch.from_pairs(pairs)
    got = _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    assert got.results["status"].tolist() == [0, _ffi.ERR_EMPTY_SEQUENCE, 0, _ffi.ERR_CODE_OUT_OF_RANGE,
                                              _ffi.ERR_NO_POSITIVE_CELL, 0]


# ---------------------------------------------------------------- error behaviour of the drop-in API
def test_api_errors(blosum62):
    with pytest.raises(AlignerError) as e:
        SimpleLocalAligner.from_str_seqs("HEAG", "PAW").perform_alignment(11.0, 2.0, blosum62,
                                                                           Heuristics(1.0, 1.0, np.ones(24)))
    assert e.value.kind == ErrorKind.UnnecessaryArgument
    with pytest.raises(AlignerError) as e:
        SimpleLocalAligner.from_str_seqs("HEAG-", "PAW")
    assert e.value.kind == ErrorKind.CharIsNotMatchable
    with pytest.raises(ReferencePanic):
        SimpleGlobalAligner.from_seqs([], [1, 2]).perform_alignment(11.0, 2.0, blosum62)
    with pytest.raises(ReferencePanic):
        SimpleLocalAligner.from_str_seqs("QQQ", "CCC").perform_alignment(11.0, 2.0, blosum62)


def test_dna_alphabet_pair(orc):
    a = SimpleLocalAligner.from_str_seqs("ATCGGATTACAGATTACA", "TTGATTACAGTTTACA", alphabet=DNA)
    r = a.perform_alignment(10.0, 1.0, nucleotide_matrix())
    ref = orc.align(orc.CORE_LOCAL, a.query, a.target, 10, 1, nucleotide_matrix())
    assert r.alignment.f == ref["f"] and r.alignment.coords == ref["coords"]
    assert r.alignment.query.tolist() == ref["qa"].tolist()


# ---------------------------------------------------------------- full size (BASELINE C4)
def test_c4_10k_pair_matches_oracle(orc, blosum62):
    """10k x 10k protein, core local 11/2, homolog variant (long real alignment): score, coords, strings."""
    q, t = workloads.c4_pair(homolog=True)
    check_pair(orc, _ffi.CORE_LOCAL, q, t, 11, 2, blosum62, full=False)


def test_c4_uniform_pair_the_bench_times(orc, blosum62):
    """The uniform-random 10k x 10k pair bench.py times as "single_pair" (its score, 8806, is quoted in the bench line)."""
    q, t = workloads.c4_pair(homolog=False)
    res = check_pair(orc, _ffi.CORE_LOCAL, q, t, 11, 2, blosum62, full=False)
    assert res.score == 8806.0 and res.flags & 2


def test_single_pair_route_repairs_the_row1_hazard_locally(orc, blosum62, monkeypatch):
    """About half of all large pairs turn out to need a different row-1 advice in a few leading columns (core local, del != ext).
    The strip pipeline then re-runs the leading columns of its first strips (passes bit 8) instead of running a second time:
    summaries and strings against the oracle over several seeds, uniform and homolog, R = 1 and R = 2, with the full direction
    matrix for the smaller ones; the same pairs with the repair switched off (full second pass) give the same answers."""
    seen_repair = 0
    for seed in range(10):
        rng = np.random.default_rng(900 + seed)
        N, M = ((2200, 1500) if seed % 2 else (3000, 4300))            # R = 1 (16+ strips) / R = 2
        q = rng.integers(0, 20, N).astype(np.uint8)
        t = workloads.mutate(q, 50 + seed, 20, 0.10, 0.02, out_len=M) if seed % 3 == 0 else rng.integers(0, 20, M).astype(np.uint8)
        res = check_pair(orc, _ffi.CORE_LOCAL, q, t, 11, 2, blosum62, full=(seed % 2 == 1), directions_only=True)
        assert res.flags & 2
        repaired = bool(res.passes & 0x100)
        seen_repair += repaired
        assert (res.passes & 0x7f) == 1 or not repaired
        if repaired:
            monkeypatch.setenv("ALN_NO_SINGLE_REPAIR", "1")
            again = check_pair(orc, _ffi.CORE_LOCAL, q, t, 11, 2, blosum62, full=False)
            monkeypatch.delenv("ALN_NO_SINGLE_REPAIR")
            assert (again.passes & 0x7f) == 2 and again.score == res.score
            # a repair run that is declared failed arms pass 1 (device-side): same answer after two passes
            monkeypatch.setenv("ALN_TEST_FAIL_REPAIR", "1")
            again = check_pair(orc, _ffi.CORE_LOCAL, q, t, 11, 2, blosum62, full=False)
            monkeypatch.delenv("ALN_TEST_FAIL_REPAIR")
            assert (again.passes & 0x7f) == 2 and not (again.passes & 0x100)
    assert seen_repair >= 2, seen_repair


def _same_batch_results(a, b, n, strings_every=1):
    assert (a.results["status"] == b.results["status"]).all()
    for f in ("score", "f", "end_y", "end_x", "start_y", "start_x", "aln_len"):
        assert (a.results[f] == b.results[f]).all(), f
    for i in range(0, n, strings_every):
        qa, ta = a.aligned(i)
        qb, tb = b.aligned(i)
        assert (qa == qb).all() and (ta == tb).all(), i


@pytest.mark.parametrize("sem", [_ffi.CORE_LOCAL, _ffi.CORE_GLOBAL])
def test_dyadic_schemes_run_on_the_integer_kernels(orc, blosum62, sem):
    """A real-valued scheme whose numbers are all multiples of 2^-k (BLOSUM62 x 0.5 with 11.5 / 2.25: k = 2; x 0.125 with 1.375 /
    0.25: k = 3) is an integer scheme scaled by 2^k, exactly: the library fills it with the integer kernels (flags bit 0) and scales
    the two scores of every summary back.  Same summaries and strings as the oracle on the real-valued scheme and as the f64
    kernels (force_f64); a scheme that is not dyadic (x 0.3) stays on the f64 kernels; with the H output the scheme is not scaled."""
    rng = np.random.default_rng(4 + sem)
    pairs = []
    for i in range(40):
        N, M = int(rng.integers(30, 1400)), int(rng.integers(30, 1400))
        q = rng.integers(0, 20, N).astype(np.uint8)
        t = rng.integers(0, 20, M).astype(np.uint8)
        L = min(N, M) // 2
        t[:L] = q[:L]
        pairs.append((q, t))
    pb = PairBatch.from_pairs(pairs)
    for scale, de, ex in ((0.5, 11.5, 2.25), (0.125, 1.375, 0.25)):
        S = blosum62 * scale
        got = _check_batch(orc, pb, sem, de, ex, S)
        assert (got.results["flags"] & 1).all(), "not on the integer kernels"
        f64 = align_batch(pb, sem, de, ex, S, force_f64=True)
        assert not (f64.results["flags"] & 1).any()
        _same_batch_results(got, f64, len(pb))
        res = check_pair(orc, sem, pairs[0][0], pairs[0][1], de, ex, S, directions_only=True)      # one pair, directions too
        assert res.flags & 1
        res = check_pair(orc, sem, pairs[1][0][:200], pairs[1][1][:150], de, ex, S)                 # with H: the f64 kernels, unscaled
        assert not (res.flags & 1)
    got = _check_batch(orc, pb, sem, 11.3, 2.1, blosum62 * 0.3)
    assert not (got.results["flags"] & 1).any()


def test_two_short_protein_pairs_per_wave(orc, blosum62):
    """The same kernel with a 24-letter alphabet: the profiles of four waves and the staged queries fit a workgroup's LDS share up to
    128 rows (R <= 4: four profile bytes per lane and code), so peptide batches run two pairs per wave too; beyond, one per wave."""
    rng = np.random.default_rng(99)
    for max_rows in (128, 200):
        pairs = []
        for i in range(3500):
            N, M = int(rng.integers(5, 300)), int(rng.integers(5, max_rows + 1))
            q = rng.integers(0, 20, N).astype(np.uint8)
            t = rng.integers(0, 20, M).astype(np.uint8)
            L = min(N, M) // 2
            t[:L] = q[:L]
            pairs.append((q, t))
        _check_batch(orc, PairBatch.from_pairs(pairs), _ffi.CORE_GLOBAL, 11, 2, blosum62)


def test_two_short_pairs_per_wave(orc, monkeypatch):
    """Core-global batches of more pairs than resident waves whose pairs all have at most 256 rows and 1024 columns are filled two
    pairs per wave (aln_fill_duo_kernel: lanes 0..31 one pair, lanes 32..63 another, each pair's directions in its own region in
    the uniform layout).  An odd number of pairs with every shape in 1..1024 x 1..256 -- so the two halves of a wave differ in both
    lengths and in the rows per lane they would pick alone --, pairs the reference would panic on in either half, score only and
    with strings: every summary against the oracle and against the one-pair-per-wave kernels (ALN_NO_DUO), strings of every
    fifth pair against the oracle."""
    rng = np.random.default_rng(2222)
    S = nucleotide_matrix()
    pairs = []
    for i in range(6001):
        N = int(rng.integers(1, 1025)) if i % 9 else int(rng.integers(1, 40))
        M = int(rng.integers(1, 257)) if i % 11 else int(rng.integers(1, 8))
        q = rng.integers(0, 4, N).astype(np.uint8)
        t = rng.integers(0, 4, M).astype(np.uint8)
        L = min(N, M) // 2
        t[:L] = q[:L]
        pairs.append((q, t))
    pairs[100] = (pairs[100][0], np.full(17, 7, np.uint8))                 # codes outside the matrix, first and second half of a wave
    pairs[4321] = (np.full(5, 250, np.uint8), pairs[4321][1])
    pairs[2000] = (pairs[2000][0], np.zeros(0, np.uint8))                  # empty sequences (the reference panics: status 2)
    pairs[2501] = (np.zeros(0, np.uint8), pairs[2501][1])
    pb = PairBatch.from_pairs(pairs)
    got = align_batch(pb, _ffi.CORE_GLOBAL, 10, 1, S)
    monkeypatch.setenv("ALN_NO_DUO", "1")
    solo = align_batch(pb, _ffi.CORE_GLOBAL, 10, 1, S)
    monkeypatch.delenv("ALN_NO_DUO")
    _same_batch_results(got, solo, len(pb), strings_every=3)
    score_only = align_batch(pb, _ffi.CORE_GLOBAL, 10, 1, S, want_traceback=False)
    assert (score_only.results["score"] == got.results["score"]).all() and (score_only.results["status"] == got.results["status"]).all()
    ref, tb, tb_off = orc.align_batch(_ffi.CORE_GLOBAL, pb.seqs, pb.q_off, pb.q_len, pb.t_off, pb.t_len, 10, 1, S, 16)
    assert ref[100].status != 0 and ref[4321].status != 0 and ref[2000].status != 0 and ref[2501].status != 0
    for i in range(len(pb)):
        r, g = ref[i], got.results[i]
        assert int(g["status"]) == r.status, i
        if r.status:
            continue
        assert (g["score"], g["f"], g["end_y"], g["end_x"], g["start_y"], g["start_x"], g["aln_len"]) == \
            (r.score, r.f, r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len), i
        if i % 5 == 0:
            cap = int(pb.q_len[i] + pb.t_len[i]) + 2
            o = int(tb_off[i])
            qa, ta = got.aligned(i)
            assert (qa == tb[o:o + r.aln_len]).all() and (ta == tb[o + cap:o + cap + r.aln_len]).all(), i


@pytest.mark.parametrize("claim", ["2", "3", "4"])
def test_waves_take_runs_of_queue_positions(orc, monkeypatch, claim):
    """Batches of many short, alike pairs (read pairs) let a wave take a run of queue positions per atomic (FillArgs::claim; C3 takes
    runs of 2 by itself).  Forced to 2 / 3 / 4 on a batch whose size is no multiple of any of them, with pairs of every length
    20..220 and two pairs the reference would panic on inside runs: every pair once, every result the oracle's."""
    rng = np.random.default_rng(5 + int(claim))
    S = nucleotide_matrix()
    pairs = []
    for i in range(7001):
        N, M = int(rng.integers(20, 221)), int(rng.integers(20, 221))
        q = rng.integers(0, 4, N).astype(np.uint8)
        t = rng.integers(0, 4, M).astype(np.uint8)
        L = min(N, M) // 2
        t[:L] = q[:L]
        pairs.append((q, t))
    pairs[1234] = (pairs[1234][0], np.full(50, 9, np.uint8))              # a code outside the matrix
    pairs[7000] = (np.full(33, 200, np.uint8), pairs[7000][1])
    pb = PairBatch.from_pairs(pairs)
    monkeypatch.setenv("ALN_CLAIM", claim)
    got = align_batch(pb, _ffi.CORE_GLOBAL, 10, 1, S)
    monkeypatch.delenv("ALN_CLAIM")
    ref, tb, tb_off = orc.align_batch(_ffi.CORE_GLOBAL, pb.seqs, pb.q_off, pb.q_len, pb.t_off, pb.t_len, 10, 1, S, 16)
    assert int(got.results["status"][1234]) == ref[1234].status != 0 and int(got.results["status"][7000]) == ref[7000].status != 0
    for i in range(len(pb)):
        r, g = ref[i], got.results[i]
        assert int(g["status"]) == r.status, i
        if r.status:
            continue
        assert (g["score"], g["f"], g["aln_len"]) == (r.score, r.f, r.aln_len), i
        if i % 7 == 0:
            cap = int(pb.q_len[i] + pb.t_len[i]) + 2
            o = int(tb_off[i])
            qa, ta = got.aligned(i)
            assert (qa == tb[o:o + r.aln_len]).all() and (ta == tb[o + cap:o + cap + r.aln_len]).all(), i


@pytest.mark.parametrize("sem", [_ffi.CORE_LOCAL, _ffi.CORE_GLOBAL, _ffi.LEGACY_GLOBAL, _ffi.LEGACY_LOCAL])
def test_wave_walk_equals_lane_walk_and_oracle(orc, blosum62, monkeypatch, sem):
    """Batches of up to 2048 pairs are walked by one WAVE per pair, whose lanes fetch the direction quads ahead of the path along
    the diagonal (tb_walk_pair_wave); larger ones by one lane per pair.  Same strings either way, and the oracle's: pairs whose
    paths run through long gaps (inserted / deleted blocks of 5..400 residues carry the path out of the fetched window, so the
    window is re-centred many times), paths that cross several strips, short last strips (every R), end cells on the borders
    (global semantics: the border runs are written by all lanes), pairs of one row or one column."""
    rng = np.random.default_rng(77 + sem)
    pairs = []
    for i in range(160):
        N = int(rng.integers(1, 2600)) if i % 7 else int(rng.integers(1, 5))
        q = rng.integers(0, 20, N).astype(np.uint8)
        t = q.copy()
        for _ in range(int(rng.integers(0, 6))):                 # block indels
            L = int(rng.integers(5, 400)); pos = int(rng.integers(0, len(t) + 1))
            if rng.random() < 0.5:
                t = np.concatenate([t[:pos], rng.integers(0, 20, L).astype(np.uint8), t[pos:]])
            else:
                t = np.concatenate([t[:pos], t[pos + L:]])
        if len(t) == 0 or i % 11 == 0:
            t = rng.integers(0, 20, int(rng.integers(1, 1500))).astype(np.uint8)
        sub = rng.random(len(t)) < 0.05
        t = np.where(sub, rng.integers(0, 20, len(t)), t).astype(np.uint8)
        pairs.append((q, t))
    pb = PairBatch.from_pairs(pairs)
    dele, ext = (11, 2) if sem in (_ffi.CORE_LOCAL, _ffi.CORE_GLOBAL) else (4, 4)
    monkeypatch.setenv("ALN_TB_WAVE", "1")
    wave = align_batch(pb, sem, dele, ext, blosum62)
    monkeypatch.setenv("ALN_TB_WAVE", "0")
    lanew = align_batch(pb, sem, dele, ext, blosum62)
    monkeypatch.delenv("ALN_TB_WAVE")
    _same_batch_results(wave, lanew, len(pb))
    ref, tb, tb_off = orc.align_batch(sem, pb.seqs, pb.q_off, pb.q_len, pb.t_off, pb.t_len, dele, ext, blosum62, 16)
    for i in range(len(pb)):
        r, g = ref[i], wave.results[i]
        assert int(g["status"]) == r.status, i
        if r.status:
            continue
        assert (g["score"], g["f"], g["end_y"], g["end_x"], g["start_y"], g["start_x"], g["aln_len"]) == \
            (r.score, r.f, r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len), i
        cap = int(pb.q_len[i] + pb.t_len[i]) + 2
        o = int(tb_off[i])
        qa, ta = wave.aligned(i)
        assert (qa == tb[o:o + r.aln_len]).all() and (ta == tb[o + cap:o + cap + r.aln_len]).all(), i


def test_batch_of_large_pairs_shares_the_strips_of_a_pair(orc, blosum62, monkeypatch):
    """256 pairs of 4200 x 4200 (1.8e7 cells each: above the old 2^24-cell line every one of them took the single-pair route, one
    after the other).  The batch kernel now takes them -- fewer pairs than resident waves, so the waves without a pair claim strips
    of other waves' pairs (cooperative passes, first passes included).  All 256 against the same batch filled with one wave per
    pair (ALN_NO_COOP), 16 of them against the oracle; a second, smaller batch goes the same way with hints suppressed (every
    open pass filled by its owner alone through the shared-pass code) and with every re-fill shared."""
    import threading
    b = workloads.c5_batch(n_pairs=256, lo=4200, hi=4200)
    got = align_batch(b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    assert (got.results["status"] == 0).all()
    assert ((got.results["flags"] & 2) == 0).all()                           # not the single-pair route
    monkeypatch.setenv("ALN_NO_COOP", "1")
    monkeypatch.setenv("ALN_BIG_TO_SINGLE", "0")
    solo = align_batch(b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    monkeypatch.delenv("ALN_NO_COOP")
    monkeypatch.delenv("ALN_BIG_TO_SINGLE")
    _same_batch_results(got, solo, len(b))
    sample = list(range(0, 256, 16))
    refs = [None] * len(sample)

    def run(j):
        i = sample[j]
        refs[j] = orc.align(orc.CORE_LOCAL, b.query(i), b.target(i), 11, 2, blosum62)
    th = [threading.Thread(target=run, args=(j,)) for j in range(len(sample))]
    [x.start() for x in th]
    [x.join() for x in th]
    for j, i in enumerate(sample):
        r, g = refs[j], got.results[i]
        assert (g["score"], g["end_y"], g["end_x"], g["start_y"], g["start_x"]) == (r["score"], r["end"][0], r["end"][1], r["start"][0], r["start"][1]), i
        qa, ta = got.aligned(i)
        assert qa.tolist() == r["qa"].tolist() and ta.tolist() == r["ta"].tolist(), i
    # the shared-pass code with nobody to share with, and with every re-fill shared whatever the queue holds
    small = workloads.c5_batch(n_pairs=1500, lo=200, hi=2000)
    base = _check_batch(orc, small, _ffi.CORE_LOCAL, 11, 2, blosum62)
    for dbg in ("2", "8"):
        monkeypatch.setenv("ALN_COOP_DEBUG", dbg)
        _same_batch_results(align_batch(small, _ffi.CORE_LOCAL, 11, 2, blosum62), base, len(small), strings_every=7)
    monkeypatch.delenv("ALN_COOP_DEBUG")
    # core global has no re-fills, only first passes to share
    _check_batch(orc, workloads.c5_batch(n_pairs=400, lo=600, hi=1800), _ffi.CORE_GLOBAL, 11, 2, blosum62)


def test_long_pairs_beyond_the_old_column_limit(orc, blosum62):
    """The single-pair route stages the whole query's profile offsets in LDS.  Up to r01 the host refused it above a 64 KiB
    budget (N ~ 29 400 columns) and such a pair fell to the one-wave batch kernel; a workgroup now opts in to the CU's whole
    160 KiB.  Bit-exact summaries and strings for 40 000 x 10 000 and 10 000 x 40 000 and at the old boundary +- 64, each on
    the strip-pipelined route (flags bit 1); the oracle runs (dense f64 H, ~0.5 min each) go side by side on host threads."""
    import threading
    rng = np.random.default_rng(4040)
    shapes = [(40000, 10000), (10000, 40000), (29376 - 64, 4160), (29376, 4160), (29376 + 64, 4160)]
    cases = []
    for N, M in shapes:
        q = rng.integers(0, 20, N).astype(np.uint8)
        # a homolog of the query's head keeps a long real alignment in the picture
        t = workloads.mutate(q[:M] if M <= N else np.concatenate([q, rng.integers(0, 20, M - N).astype(np.uint8)]),
                             1234 + N, 20, 0.10, 0.02, out_len=M)
        cases.append((q, t))
    refs = [None] * len(cases)

    def run(i):
        refs[i] = orc.align(orc.CORE_LOCAL, cases[i][0], cases[i][1], 11, 2, blosum62)
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(cases))]
    [x.start() for x in th]
    got = []
    for q, t in cases:
        got.append(runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, blosum62))
        # the route, not the clock, is what this test asserts (bench.py's single_pair.long_pairs reports the rate: >= 50 GCUPS)
        assert got[-1][0].flags & 2, "not on the strip-pipelined route"
    [x.join() for x in th]
    for (res, qa, ta, _, _), ref in zip(got, refs):
        assert res.status == 0 and ref["status"] == 0
        assert (res.score, res.f, (res.end_y, res.end_x), (res.start_y, res.start_x)) == (ref["score"], ref["f"], ref["end"], ref["start"])
        assert qa.tolist() == ref["qa"].tolist() and ta.tolist() == ref["ta"].tolist()


@pytest.mark.parametrize("sem", [_ffi.CORE_LOCAL, _ffi.CORE_GLOBAL])
def test_pairs_beyond_the_single_route_column_limit(orc, blosum62, sem):
    """Above ~77 000 columns the single-pair route cannot stage the query's profile offsets in LDS and the pair stays in the
    batch kernel (aln_host.hip, routing); the reference's only limit is memory (simple/mod.rs:53-57).  110 000 columns x
    700 rows (two strips of the batch layout), its transpose (1719 strips: the single-pair route again), and 131 200 x 130
    (one strip), alone and as a three-pair batch: summaries and both strings against the oracle."""
    rng = np.random.default_rng(7707)
    pairs = []
    for N, M in [(110000, 700), (700, 110000), (131200, 130)]:
        q = rng.integers(0, 20, N).astype(np.uint8)
        t = workloads.mutate(q[:M] if M <= N else np.concatenate([q, rng.integers(0, 20, M - N).astype(np.uint8)]),
                             99 + N, 20, 0.10, 0.02, out_len=M)
        pairs.append((q, t))
    for q, t in pairs:
        ref = orc.align(sem, q, t, 11, 2, blosum62)
        res, qa, ta, _, _ = runtime.align_pair(sem, q, t, 11, 2, blosum62)
        assert res.status == 0 and ref["status"] == 0
        assert (res.score, res.f, (res.end_y, res.end_x), (res.start_y, res.start_x)) == (ref["score"], ref["f"], ref["end"], ref["start"])
        assert qa.tolist() == ref["qa"].tolist() and ta.tolist() == ref["ta"].tolist()
    b = PairBatch.from_pairs(pairs)
    _check_batch(orc, b, sem, 11, 2, blosum62)


def test_c5_sample_matches_oracle(orc, blosum62):
    """BASELINE C5 at its real lengths (200..2000 aa) on a 1500-pair sample: every summary and both aligned strings.
    Exercises multi-strip pairs, the checkpointed first pass and the localized row-1 repair."""
    b = workloads.c5_batch(n_pairs=1500)
    got = _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    passes = got.results["passes"]
    assert ((passes & 0x80) == 0).all()              # nobody needed the strict-order routine
    assert ((passes & 0xff) == 1).mean() > 0.9       # the first full pass (+ at most a localized repair) suffices


def test_c3_full_batch_properties():
    """BASELINE C3 at full size (10 000 read pairs, core global +5/-4, 10/1): size-independent properties --
    every pair OK, global coords, score == recomputed score of the returned alignment, strings consistent."""
    b = workloads.c3_batch(10000)
    S = nucleotide_matrix()
    got = align_batch(b, _ffi.CORE_GLOBAL, 10, 1, S)
    r = got.results
    assert (r["status"] == 0).all() and (r["f"] == 0.0).all()
    assert (r["end_y"] == 150).all() and (r["end_x"] == 150).all() and (r["start_y"] == 0).all() and (r["start_x"] == 0).all()
    for i in range(0, 10000, 97):
        qa, ta = got.aligned(i)
        # the duplicated seed pair closes both strings (simple/mod.rs:102-105)
        assert qa[-1] == b.query(i)[-1] and ta[-1] == b.target(i)[-1]
        body_q, body_t = qa[:-1], ta[:-1]
        assert (body_q[body_q != 98] == b.query(i)).all() and (body_t[body_t != 98] == b.target(i)).all()
        assert not ((body_q == 98) & (body_t == 98)).any()


def test_summary_gather_on_gpu(blosum62):
    """The records bench.py gathers with RCCL: device pointer -> torch tensor (no copy) -> all_gather (world size 1)."""
    import os
    import torch
    import torch.distributed as dist
    from aligner_amd.batch import RESULT_DTYPE, StagedBatch
    from aligner_amd.distributed import SummaryGather, device_bytes_as_tensor, lpt_shards
    b = workloads.c5_batch(n_pairs=200, lo=30, hi=200)
    sb = StagedBatch(b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        sb.run(stream.cuda_stream)
    torch.cuda.synchronize()
    host = sb.fetch(False).results
    view = device_bytes_as_tensor(sb.results_device_ptr, len(b) * RESULT_DTYPE.itemsize)
    assert (view.cpu().numpy().view(RESULT_DTYPE) == host).all()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        shards = lpt_shards(b.q_len.astype(np.int64) * b.t_len.astype(np.int64), 1)
        g = SummaryGather([len(s) for s in shards], 0, "cuda")
        with torch.cuda.stream(stream):
            g(view)
        torch.cuda.synchronize()
        assert (g.unpack(shards, len(b)) == host).all()
    finally:
        dist.destroy_process_group()
    sb.close()


# ---------------------------------------------------------------- callers either side of the path
def test_score_only_batch_equals_full_batch(blosum62):
    b = workloads.c5_batch(n_pairs=400, lo=50, hi=900)
    full = align_batch(b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    score_only = align_batch(b, _ffi.CORE_LOCAL, 11, 2, blosum62, want_traceback=False)
    for k in ("f", "score", "end_y", "end_x", "status"):
        assert (full.results[k] == score_only.results[k]).all(), k


def test_p_value_batch_driver(orc, blosum62):
    """calculate_p_value's batch (statistics/mod.rs:255-286): 4 999 shuffled targets, score only, one GPU batch; the
    scores equal the oracle's on the same shuffles."""
    from aligner_amd import statistics
    rng = np.random.default_rng(2024)
    q = rng.integers(0, 20, 180).astype(np.uint8)
    t = np.concatenate([q[20:150], rng.integers(0, 20, 60).astype(np.uint8)])
    init = orc.align(orc.CORE_LOCAL, q, t, 11, 2, blosum62)["f"]
    scores, lengths, batch = statistics.shuffled_scores(q, t, init, 11, 2, blosum62, rng=np.random.default_rng(7))
    assert len(scores) == len(lengths) == 5000 and len(batch) == 4999
    assert scores[0] == init and lengths[0] == len(t) and (lengths[1:] >= len(t) - 6).all()
    ref, _, _ = orc.align_batch(orc.CORE_LOCAL, batch.seqs, batch.q_off, batch.q_len, batch.t_off, batch.t_len, 11, 2,
                                blosum62, n_threads=8, want_traceback=False)
    assert all(ref[i].f == scores[i + 1] for i in range(len(batch)))
    p = statistics.calculate_p_value(q, t, init, 11, 2, blosum62, rng=np.random.default_rng(7))
    assert 0.0 <= p <= 1.0 and p < 0.05          # a 130-residue exact match is not a chance hit


def test_cli_on_reference_example(orc, blosum62, capsys, tmp_path):
    import os
    from aligner_amd import cli
    from aligner_amd.fasta import encode_records, read_fasta
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "protein.fasta")
    q, t = encode_records(read_fasta(path), Protein)
    for flag, sem in (([], orc.CORE_LOCAL), (["-g"], orc.CORE_GLOBAL)):
        assert cli.main(["-i", path] + flag) == 0
        out = capsys.readouterr().out.strip()
        ref = orc.align(sem, q, t, 11, 2, blosum62)
        assert out == cli.debug_vec(orc.midline(ref["qa"], ref["ta"], blosum62))
    # SURVEY Appendix B anchor: f = 1554 on this pair
    assert orc.align(orc.CORE_LOCAL, q, t, 11, 2, blosum62)["f"] == 1554.0
    # C1 plumbing: the book example through the CLI, both modes (SURVEY Appendix B anchors)
    book = os.path.join(os.path.dirname(path), "book_example_1.fasta")
    assert cli.main(["-i", book]) == 0
    assert capsys.readouterr().out.strip() == "[Blank, A, W, Blank, H, E, Blank, E, E]"
    assert cli.main(["-i", book, "-g"]) == 0
    assert capsys.readouterr().out.strip() == "[Blank, Blank, A, Blank, Blank, W, Blank, H, E, Blank, E, E]"


# ---------------------------------------------------------------- PWM aligner
def _check_pwm(orc, seq, dele, ext, pwm, full=True, **kw):
    from aligner_amd.pwm import PWMAligner
    ref = orc.align_pwm(seq, dele, ext, pwm, want_matrices=full)
    r = PWMAligner.from_seqs(seq).perform_alignment(dele, ext, pwm, want_matrices=full, **kw)
    a = r.alignment
    assert a.f == ref["f"] and a.coords == ref["coords"], (a.f, ref["f"], a.coords, ref["coords"])
    assert a.numbered.tolist() == ref["numbered"].tolist() and a.query.tolist() == ref["qal"].tolist()
    if full:
        assert (r.alignment_matrix == ref["H"]).all() and (r.direction_matrix == ref["D"]).all()
    return r


@pytest.mark.parametrize("shape", [(40, 30), (330, 300), (700, 300), (64, 1), (5, 600)])
def test_pwm_aligner_matches_oracle(orc, shape):
    """PWMAligner::perform_alignment (pwm/mod.rs:29-126): random -1/0/1 PWMs (lib.rs:92-96) and a planted motif."""
    Q, W = shape
    rng = np.random.default_rng(Q * 7 + W)
    pwm = rng.integers(-1, 2, (4, W)).astype(np.float64)
    seq = rng.integers(0, 4, Q).astype(np.uint8)
    for dele, ext in ((3, 1), (2, 2), (1, 2)):
        _check_pwm(orc, seq, dele, ext, pwm)                       # generic kernels (H dump)
        _check_pwm(orc, seq, dele, ext, pwm, full=False)           # fast kernels (v_perm byte select)
    motif = rng.integers(0, 4, W).astype(np.uint8)
    pwm2 = -np.ones((4, W)); pwm2[motif, np.arange(W)] = 2.0
    if Q > W + 10:
        seq2 = seq.copy(); seq2[5:5 + W] = motif
        r = _check_pwm(orc, seq2, 3, 1, pwm2, full=False)
        assert r.alignment.f == 2.0 * W
    # real-valued PWM -> f64 kernels
    _check_pwm(orc, seq, 1.5, 0.4, np.round(rng.normal(0, 1, (4, W)), 2), full=False)


def test_pwm_errors_and_empty_result(orc):
    from aligner_amd.pwm import PWMAligner
    seq = np.array([0, 1, 2, 3, 0, 1], np.uint8)
    with pytest.raises(AlignerError) as e:
        PWMAligner.from_seqs(seq).perform_alignment(3, 1, np.zeros((3, 5)))
    assert e.value.kind == ErrorKind.MatrixShapeError
    with pytest.raises(AlignerError) as e:
        PWMAligner.from_seqs(seq).perform_alignment(3, 1, np.zeros((4, 5)), Heuristics(1, 1, np.ones(4)))
    assert e.value.kind == ErrorKind.UnnecessaryArgument
    r = PWMAligner.from_seqs(seq).perform_alignment(3, 1, -np.ones((4, 5)))       # no positive cell: empty, no panic
    assert r.alignment.f == 0.0 and len(r.alignment.numbered) == 0 and r.alignment.coords == ((1, 1), (1, 1))
    ref = orc.align_pwm(seq, 3, 1, -np.ones((4, 5)))
    assert ref["f"] == 0.0 and ref["coords"] == ((1, 1), (1, 1))


def test_pwm_empty_sequence_is_ok_and_empty(orc):
    """PWMAligner with an empty sequence: the reference's loops do not run, argmax is (0, 0), the traceback stops at once --
    Ok with an empty alignment and f = 0 (pwm/mod.rs:52-108), not a panic; alone and inside a window batch."""
    from aligner_amd.pwm import PWMAligner, align_windows
    pwm = np.arange(20, dtype=np.float64).reshape(4, 5) - 6
    r = PWMAligner.from_seqs(np.zeros(0, np.uint8)).perform_alignment(3, 1, pwm)
    ref = orc.align_pwm(np.zeros(0, np.uint8), 3, 1, pwm)
    assert ref["status"] == 0 and ref["f"] == 0.0 and len(ref["numbered"]) == 0
    assert r.alignment.f == 0.0 and len(r.alignment.numbered) == 0 and r.alignment.coords == ref["coords"]
    res, alns = align_windows([np.array([0, 1, 2], np.uint8), np.zeros(0, np.uint8), np.array([3, 3, 1, 0], np.uint8)], 3, 1, pwm)
    assert res["status"].tolist() == [0, 0, 0] and res["aln_len"][1] == 0 and res["f"][1] == 0.0
    for i, w in enumerate(([0, 1, 2], [], [3, 3, 1, 0])):
        ref = orc.align_pwm(np.array(w, np.uint8), 3, 1, pwm)
        assert res["f"][i] == ref["f"] and alns[i].numbered.tolist() == ref["numbered"].tolist()


def test_larger_alphabets_and_strided_matrices(orc):
    """Alphabets of 28 letters (S and four waves' query profiles still fit 64 KiB of LDS: fast kernels) and of 40 and 64 letters
    (they do not: generic kernels) against the oracle; and a matrix handed over as a broadcast / transposed numpy view
    (row stride 0 or not a multiple of the row length) is compacted on the way in."""
    rng = np.random.default_rng(64)
    for A in (28, 40, 64):
        S = rng.integers(-6, 9, (A, A)).astype(np.float64)
        q = rng.integers(0, A, 300).astype(np.uint8)
        t = rng.integers(0, A, 700).astype(np.uint8)
        for sem in (_ffi.CORE_LOCAL, _ffi.CORE_GLOBAL, _ffi.LEGACY_LOCAL):
            res = check_pair(orc, sem, q, t, 7, 7 if sem == _ffi.LEGACY_LOCAL else 2, S, full=True, directions_only=True)
            assert res.flags & 1
        b = PairBatch.from_pairs([(rng.integers(0, A, int(rng.integers(5, 400))).astype(np.uint8),
                                   rng.integers(0, A, int(rng.integers(5, 900))).astype(np.uint8)) for _ in range(40)])
        _check_batch(orc, b, _ffi.CORE_LOCAL, 7, 2, S)
    base = rng.integers(-4, 6, 24).astype(np.float64)
    views = [np.broadcast_to(base, (24, 24)), np.asfortranarray(rng.integers(-4, 6, (24, 24)).astype(np.float64)),
             rng.integers(-4, 6, (24, 48)).astype(np.float64)[:, ::2]]
    q = rng.integers(0, 20, 90).astype(np.uint8)
    t = rng.integers(0, 20, 130).astype(np.uint8)
    for S in views:
        check_pair(orc, _ffi.CORE_LOCAL, q, t, 11, 2, np.array(S), full=False)
        res, qa, ta, _, _ = runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, S)
        ref = orc.align(orc.CORE_LOCAL, q, t, 11, 2, np.ascontiguousarray(S))
        assert res.score == ref["score"] and qa.tolist() == ref["qa"].tolist()


def test_pwm_window_batch(orc):
    """latent-repeat-search's inner loop as one batch: 400 windows of 330 nt against one 300-column PWM."""
    from aligner_amd.pwm import align_windows
    rng = np.random.default_rng(99)
    pwm = rng.integers(-1, 2, (4, 300)).astype(np.float64)
    chrom = rng.integers(0, 4, 20000).astype(np.uint8)
    wins = [chrom[i * 45:i * 45 + 330] for i in range(400)]
    res, alns = align_windows(wins, 3, 1, pwm)
    for i in range(0, 400, 7):
        ref = orc.align_pwm(wins[i], 3, 1, pwm)
        assert res["f"][i] == ref["f"] and alns[i].coords == ref["coords"]
        assert alns[i].numbered.tolist() == ref["numbered"].tolist() and alns[i].query.tolist() == ref["qal"].tolist()
        fm = alns[i].get_frequency_matrix()
        assert fm.shape == (4, 300) and fm.sum() == ((ref["numbered"] != 0) & (ref["qal"] != 98)).sum()


# ---------------------------------------------------------------- heuristic re-estimation loop
def test_heuristic_aligner_loop_matches_oracle_loop(orc, blosum62):
    """HeuristicAligner (heuristic/mod.rs:36-78): every iteration is a core-local alignment with a real-valued matrix
    (f64 kernels).  The same loop driven by the CPU oracle must visit the same matrices and end on the same result."""
    from aligner_amd.alignment import Alignment
    from aligner_amd.heuristic import HeuristicAligner, transform_matrix
    rng = np.random.default_rng(31)
    q = rng.integers(0, 20, 160).astype(np.uint8)
    t = np.concatenate([rng.integers(0, 20, 30).astype(np.uint8), q[40:130], rng.integers(0, 20, 25).astype(np.uint8)])
    freqs = np.bincount(t, minlength=24).astype(np.float64) / len(t)
    h = Heuristics(kd=-0.5, r_squared=0.0, frequencies=freqs)
    got = HeuristicAligner.from_seqs(q, t, Protein).perform_alignment(11.0, 2.0, blosum62, h)
    # oracle-driven replay of the loop
    r2 = float(blosum62.shape[0] * blosum62.shape[1])
    m = transform_matrix(blosum62, h.kd, r2, freqs)
    max_f, iters = 0.0, 0
    while True:
        ref = orc.align(orc.CORE_LOCAL, q, t, 11.0, 2.0, m)
        assert ref["status"] == 0
        iters += 1
        if ref["f"] > max_f:
            max_f = ref["f"]
            m = transform_matrix(Alignment(Protein, ref["qa"], ref["ta"], ref["coords"], ref["f"]).get_frequency_matrix(),
                                 h.kd, r2, freqs)
        else:
            break
    assert iters >= 2
    assert got.alignment.f == ref["f"] and got.alignment.coords == ref["coords"]
    assert got.alignment.query.tolist() == ref["qa"].tolist() and got.alignment.target.tolist() == ref["ta"].tolist()
    assert (got.matrix == m).all()
    with pytest.raises(AlignerError) as e:
        HeuristicAligner.from_seqs(q, t, Protein).perform_alignment(11.0, 2.0, blosum62, None)
    assert e.value.kind == ErrorKind.MissingArgument


def test_heuristic_pwm_aligner_loop_matches_oracle_loop(orc):
    """HeuristicPWMAligner (heuristic/mod.rs:104-140): the same loop around PWMAligner, frequency matrix 4 x W."""
    from aligner_amd.enums import DNA
    from aligner_amd.heuristic import HeuristicPWMAligner, transform_matrix
    from aligner_amd.pwm import PWMAlignment
    rng = np.random.default_rng(77)
    W = 60
    unit = rng.integers(0, 4, W).astype(np.uint8)
    seq = np.concatenate([rng.integers(0, 4, 40).astype(np.uint8)] +
                         [np.where(rng.random(W) < 0.15, rng.integers(0, 4, W), unit).astype(np.uint8) for _ in range(3)])
    freqs = np.bincount(seq, minlength=4).astype(np.float64) / len(seq)
    start = rng.normal(0, 1, (4, W))
    h = Heuristics(kd=-0.3, r_squared=float(4 * W), frequencies=freqs)
    got = HeuristicPWMAligner.from_seqs(seq, None, DNA).perform_alignment(4.0, 1.0, start, h)
    m = transform_matrix(start, h.kd, h.r_squared, freqs)
    max_f, iters = 0.0, 0
    while True:
        ref = orc.align_pwm(seq, 4.0, 1.0, m)
        assert ref["status"] == 0
        iters += 1
        if ref["f"] > max_f:
            max_f = ref["f"]
            m = transform_matrix(PWMAlignment(DNA, ref["numbered"], ref["qal"], W, ref["coords"], ref["f"])
                                 .get_frequency_matrix(), h.kd, h.r_squared, freqs)
        else:
            break
    assert iters >= 2
    assert got.alignment.f == ref["f"] and got.alignment.coords == ref["coords"]
    assert got.alignment.numbered.tolist() == ref["numbered"].tolist() and got.alignment.query.tolist() == ref["qal"].tolist()
    assert (got.matrix == m).all()
    with pytest.raises(AlignerError) as e:
        HeuristicPWMAligner.from_seqs(seq, None, DNA).perform_alignment(4.0, 1.0, start, None)
    assert e.value.kind == ErrorKind.MissingArgument


# ---------------------------------------------------------------- single-pair kernel (one wave per strip, asm steady state)
@pytest.mark.parametrize("r", ["1", "2"])
@pytest.mark.parametrize("shape", [(500, 700), (1000, 300), (2100, 130), (777, 1031), (2500, 200), (4200, 129), (64, 4200), (130, 2100),
                                   (95, 2800)])
def test_single_pair_kernel_forced_rows_per_lane(orc, blosum62, monkeypatch, r, shape):
    """The strip-pipelined kernel with its hand-scheduled steady-state loop (tools/gen_single_asm.py), R = 1 and R = 2 rows
    per lane forced through ALN_SINGLE_R: every direction of the matrix, end cell, coordinates and both strings.
    Shapes: ragged last strips, N not a multiple of 64, N across the 2048-step tracker boundary and the 4096-column LDS
    ring, N too short for a single unmasked quad (every quad runs the masked loop); zero-rich +-1 scoring
    makes the row-1 hazard bite (several passes, strip 0 switching between the asm loop and the C++ step)."""
    monkeypatch.setenv("ALN_SINGLE_R", r)
    N, M = shape
    rng = np.random.default_rng(1000 * N + M)
    q = rng.integers(0, 4, N).astype(np.uint8)
    t = rng.integers(0, 4, M).astype(np.uint8)
    S4 = np.where(np.eye(4) > 0, 1.0, -1.0)
    for gaps in ((2, 1), (1, 2), (11, 2)):
        res = check_pair(orc, _ffi.CORE_LOCAL, q, t, gaps[0], gaps[1], S4, directions_only=True)
        assert res.flags & 2, "expected the single-pair route"
    qp = rng.integers(0, 20, N).astype(np.uint8)
    tp = np.concatenate([rng.integers(0, 20, M // 3).astype(np.uint8), np.tile(qp, M // N + 2)[N // 4:N // 4 + M - M // 3]])[:M]
    res = check_pair(orc, _ffi.CORE_LOCAL, qp, tp, 11, 2, blosum62, directions_only=True)
    assert res.flags & 2


@pytest.mark.parametrize("r", ["1", "2"])
def test_single_pair_kernel_one_strip_per_workgroup(orc, blosum62, monkeypatch, r):
    """The same kernel with one wave per workgroup (every hand-off through the granule rows) -- the configuration used
    when the query is too long for four strips' LDS; forced here through ALN_SINGLE_W1."""
    monkeypatch.setenv("ALN_SINGLE_R", r)
    monkeypatch.setenv("ALN_SINGLE_W1", "1")
    rng = np.random.default_rng(77)
    for N, M in ((777, 1031), (2500, 200)):
        q = rng.integers(0, 4, N).astype(np.uint8)
        t = rng.integers(0, 4, M).astype(np.uint8)
        res = check_pair(orc, _ffi.CORE_LOCAL, q, t, 2, 1, np.where(np.eye(4) > 0, 1.0, -1.0), directions_only=True)
        assert res.flags & 2
        qp = rng.integers(0, 20, N).astype(np.uint8)
        tp = np.concatenate([rng.integers(0, 20, M // 3).astype(np.uint8), np.tile(qp, M // N + 2)[N // 4:N // 4 + M - M // 3]])[:M]
        assert check_pair(orc, _ffi.CORE_LOCAL, qp, tp, 11, 2, blosum62, directions_only=True).flags & 2


@pytest.mark.parametrize("sem", SEMS)
@pytest.mark.parametrize("shape", [(2500, 100), (2498, 938), (4200, 70)])
def test_queries_longer_than_one_tracker_chunk(orc, blosum62, sem, shape):
    """N + 63 > 2048 wave steps: the fill is cut into 2048-step chunks (the local end-cell tracker is folded there).  The
    non-local semantics once never left the first chunk (an endless loop in the kernel) -- found by tools/fuzz_parity.py.
    Both routes: (2500, 100) and (4200, 70) go to the batch kernel, (2498, 938) to the single-pair kernel."""
    N, M = shape
    rng = np.random.default_rng(N + M + sem)
    q = rng.integers(0, 20, N).astype(np.uint8)
    t = rng.integers(0, 20, M).astype(np.uint8)
    gaps = (11, 2) if sem in (_ffi.CORE_GLOBAL, _ffi.CORE_LOCAL) else (8, 8)
    check_pair(orc, sem, q, t, gaps[0], gaps[1], blosum62, directions_only=True)


@pytest.mark.parametrize("drop", ["1", "6", "8"])
def test_single_pair_lost_producer_poisons_the_run(blosum62, monkeypatch, drop):
    """Fault injection (ALN_TEST_DROP_STRIP): strip `drop - 1` never runs.  Every strip below it polls a bounded number of
    times, the run comes back as a device error (never a wrong answer, never a hang), and the next call works."""
    import time
    from aligner_amd.errors import DeviceError
    rng = np.random.default_rng(1)
    q = rng.integers(0, 20, 2000).astype(np.uint8)
    t = rng.integers(0, 20, 2000).astype(np.uint8)
    monkeypatch.setenv("ALN_TEST_DROP_STRIP", drop)
    t0 = time.time()
    with pytest.raises(DeviceError):
        runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, blosum62)
    assert time.time() - t0 < 30
    monkeypatch.delenv("ALN_TEST_DROP_STRIP")
    res = runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, blosum62)[0]
    assert res.status == 0 and res.flags & 2


@pytest.mark.parametrize("r", ["1", "2"])
@pytest.mark.parametrize("shape", [(992, 300), (1024, 700), (2498, 938), (777, 1031)])
def test_single_pair_kernel_core_global(orc, blosum62, monkeypatch, r, shape):
    """The single-pair route's asm loop for the core-global semantics (six-instruction cells; strip 0 feeds the border
    H[0][x] = -x del through the group register).  N a multiple of the quad keeps the overwritten corner H[0][N] out of
    that loop (simple/mod.rs:62); every direction, the corner score and both strings."""
    monkeypatch.setenv("ALN_SINGLE_R", r)
    N, M = shape
    rng = np.random.default_rng(N * 7 + M)
    q = rng.integers(0, 20, N).astype(np.uint8)
    t = rng.integers(0, 20, M).astype(np.uint8)
    t[M // 4:M // 4 + min(N, M) // 2] = q[N // 4:N // 4 + min(N, M) // 2][:len(t[M // 4:M // 4 + min(N, M) // 2])]
    for gaps in ((11, 2), (3, 3)):
        res = check_pair(orc, _ffi.CORE_GLOBAL, q, t, gaps[0], gaps[1], blosum62, directions_only=True)
        assert res.flags & 2


@pytest.mark.parametrize("sem", [_ffi.CORE_GLOBAL, _ffi.CORE_LOCAL])
def test_single_pair_traceback_outside_the_exit_map_band(orc, blosum62, sem):
    """The exit maps of the single-pair traceback cover a band of 1024 columns around the slope-1 line through the start
    cell; a path with a 1000-column gap leaves it, and the chain kernel then walks those strips itself."""
    rng = np.random.default_rng(404)
    a = rng.integers(0, 20, 1000).astype(np.uint8)
    b = rng.integers(0, 20, 1000).astype(np.uint8)
    x = rng.integers(0, 20, 1000).astype(np.uint8)
    for q, t in ((np.concatenate([a, x, b]), np.concatenate([a, b])), (np.concatenate([a, b]), np.concatenate([a, x, b]))):
        res = check_pair(orc, sem, q, t, 11, 2, blosum62, full=False)
        assert res.flags & 2 and res.aln_len > 2900


def test_overlapped_traceback_batch(orc, blosum62, monkeypatch):
    """A batch large enough (>= 4096 pairs; ALN_TB_OVERLAP_ANY lifts the 2e9-cell floor) for the walk kernel to run beside the
    fill kernel: every summary and both strings against the oracle, and a second run of the same staged batch (new epoch of
    the "walked" marks) repeats them."""
    monkeypatch.setenv("ALN_TB_OVERLAP_ANY", "1")
    b = workloads.c5_batch(n_pairs=5000, lo=30, hi=260)
    got = _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    from aligner_amd.batch import StagedBatch
    sb = StagedBatch(b, _ffi.CORE_LOCAL, 11, 2, blosum62, outputs=_ffi.OUT_SCORE | _ffi.OUT_TRACEBACK)
    for _ in range(3):
        sb.run()
    sb.sync()
    again = sb.fetch(want_traceback=True)
    assert (again.results == got.results).all()
    for i in range(0, len(b), 97):
        qa, ta = got.aligned(i)
        qb, tb = again.aligned(i)
        assert (qa == qb).all() and (ta == tb).all()


def test_overlapped_traceback_gives_up_cleanly(orc, blosum62, monkeypatch):
    """The walk kernel that runs beside the fill may give up on a wait (ALN_TB_WAIT_US=0: at the first entry that is not
    there yet); the sweep after the fill walks what it left, and the results do not change."""
    monkeypatch.setenv("ALN_TB_OVERLAP_ANY", "1")
    monkeypatch.setenv("ALN_TB_WAIT_US", "0")
    b = workloads.c5_batch(n_pairs=4500, lo=30, hi=200)
    _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    monkeypatch.setenv("ALN_TB_OVERLAP", "0")
    _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62)


def test_pwm_window_batch_with_overlapped_traceback(orc, monkeypatch):
    """5000 short windows against one PWM: enough pairs for the walk kernel to run beside the fill (PWM tag strings sit
    behind the u32 column numbers, and a window without a positive cell has no walk at all)."""
    from aligner_amd.pwm import align_windows
    monkeypatch.setenv("ALN_TB_OVERLAP_ANY", "1")
    rng = np.random.default_rng(4242)
    pwm = rng.integers(-2, 3, (4, 48)).astype(np.float64)
    chrom = rng.integers(0, 4, 60000).astype(np.uint8)
    wins = [chrom[i * 11:i * 11 + int(rng.integers(5, 140))] for i in range(5000)]
    res, alns = align_windows(wins, 3, 1, pwm)
    for i in range(0, 5000, 3):
        ref = orc.align_pwm(wins[i], 3, 1, pwm)
        assert res["f"][i] == ref["f"] and alns[i].coords == ref["coords"], i
        assert alns[i].numbered.tolist() == ref["numbered"].tolist() and alns[i].query.tolist() == ref["qal"].tolist(), i


def test_c5_full_batch_against_oracle_digest(blosum62):
    """BASELINE C5 at FULL size (100 000 pairs, 1.2e11 cells): every score, and per block of 1000 pairs a SHA-256 over the
    summaries (score, end, start, length) and both aligned strings, against tests/golden/c5_100k_digest.npz -- generated by
    the CPU oracle (tests/golden/make_c5_golden.py, ~10 min on 8 cores), so the whole headline workload is bit-exact."""
    import hashlib
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c5_100k_digest.npz"))
    block = int(g["block"])
    b = workloads.c5_batch(100000)
    got = align_batch(b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    r = got.results
    assert (r["status"] == 0).all()
    assert (r["score"].astype(np.int64) == g["scores"]).all()
    bad = []
    for k in range(len(b) // block):
        h = hashlib.sha256()
        lo = k * block
        summ = np.stack([r[f][lo:lo + block].astype(np.int32) for f in ("score", "end_y", "end_x", "start_y", "start_x", "aln_len")], axis=1)
        h.update(np.ascontiguousarray(summ).tobytes())
        for i in range(lo, lo + block):
            qa, ta = got.aligned(i)
            h.update(np.ascontiguousarray(qa, dtype=np.uint8).tobytes())
            h.update(np.ascontiguousarray(ta, dtype=np.uint8).tobytes())
        if h.digest() != g["digests"][k].tobytes():
            bad.append(k)
    assert not bad, "blocks with a different digest: %s" % bad[:10]
    # the staged API (whole batch resident in HBM, what bench.py times) gives the same bytes as the pipelined call
    from aligner_amd.batch import align_batch_staged
    again = align_batch_staged(b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    assert (again.results == r).all()
    for i in range(0, len(b), 1013):
        assert all((x == y).all() for x, y in zip(again.aligned(i), got.aligned(i)))


def _spoil(b, rng, n_bad=7):
    """Copies of the batch's arrays with a few residue codes outside the matrix and a few empty sequences."""
    seqs = b.seqs.copy()
    q_len, t_len = b.q_len.copy(), b.t_len.copy()
    idx = rng.choice(len(b), size=2 * n_bad, replace=False)
    for j, i in enumerate(idx[:n_bad]):
        off = int(b.q_off[i] if j % 2 else b.t_off[i])
        ln = int(b.q_len[i] if j % 2 else b.t_len[i])
        seqs[off + int(rng.integers(0, ln))] = 24 + j          # outside the 24 x 24 matrix
    for j, i in enumerate(idx[n_bad:]):
        if j % 2:
            q_len[i] = 0
        else:
            t_len[i] = 0
    return PairBatch(seqs, b.q_off, q_len, b.t_off, t_len)


def test_pipelined_batch_call_chunks_layouts_and_bad_pairs(orc, blosum62, monkeypatch):
    """aln_align_batch is a pipeline over chunks of the caller's pair order (aln_host.hip).  Forced into ~15 chunks of a small
    batch (ALN_CHUNK_CELLS): every summary and both strings against the oracle for (a) the documented cumulative tb layout
    (copied back as one span per chunk), (b) a foreign layout (reversed, with gaps: staged and scattered per string),
    (c) sequences scattered over a sparse buffer (gathered into staging), (d) pairs the reference panics on (codes outside the
    matrix are found by the device-side validation; empty sequences on the host) spread over the chunks."""
    rng = np.random.default_rng(99)
    b = workloads.c5_batch(n_pairs=640, lo=20, hi=520)
    monkeypatch.setenv("ALN_CHUNK_CELLS", str(b.cells // 15))
    ref = _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    # (b) reversed offsets with gaps of 5 bytes
    cap = 2 * (b.q_len + b.t_len + np.uint64(2)) + np.uint64(5)
    off = np.zeros(len(b), dtype=np.uint64)
    off[::-1][1:] = np.cumsum(cap[::-1])[:-1]
    got = _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62, tb_off=off)
    assert (got.results == ref.results).all()
    # (c) the same pairs, every sequence at a far-away offset of a sparse buffer
    gap = 9000
    big = np.full(2 * len(b) * gap + 16, 77, dtype=np.uint8)
    q_off = (np.arange(len(b), dtype=np.uint64) * 2 + 1) * gap
    t_off = (np.arange(len(b), dtype=np.uint64) * 2) * gap + 13
    for i in range(len(b)):
        big[int(q_off[i]):int(q_off[i] + b.q_len[i])] = b.query(i)
        big[int(t_off[i]):int(t_off[i] + b.t_len[i])] = b.target(i)
    sparse = PairBatch(big, q_off, b.q_len, t_off, b.t_len)
    got = align_batch(sparse, _ffi.CORE_LOCAL, 11, 2, blosum62)
    assert (got.results == ref.results).all()
    for i in range(0, len(b), 7):
        assert all((x == y).all() for x, y in zip(got.aligned(i), ref.aligned(i)))
    # (d) bad pairs in several chunks, other semantics through the same pipeline
    bad = _spoil(b, rng)
    got = _check_batch(orc, bad, _ffi.CORE_LOCAL, 11, 2, blosum62)
    assert sorted(set(got.results["status"].tolist())) == [0, _ffi.ERR_EMPTY_SEQUENCE, _ffi.ERR_CODE_OUT_OF_RANGE]
    _check_batch(orc, bad, _ffi.CORE_GLOBAL, 11, 2, blosum62)
    _check_batch(orc, b, _ffi.LEGACY_LOCAL, 11, 11, blosum62)
    _check_batch(orc, b, _ffi.CORE_LOCAL, 11.5, 2.25, blosum62 * 0.5, force_f64=True)           # f64 kernels
    # one chunk again: the same answers
    monkeypatch.delenv("ALN_CHUNK_CELLS")
    got = _check_batch(orc, bad, _ffi.CORE_LOCAL, 11, 2, blosum62)


def test_multi_device_context_shards_chunks(orc, blosum62, monkeypatch):
    """aln_create_multi: one context over a LIST of GPUs; a batch call's chunks are taken from a common queue by one pipeline
    per device, each writing its chunks straight into the caller's buffers.  Run over every visible device and -- so that the
    sharded path also runs on a one-GPU box -- over a list naming device 0 three times (three pools, three pipelines):
    same summaries and strings as the single-device call, for standard and foreign tb layouts and with bad pairs."""
    import torch
    rng = np.random.default_rng(17)
    b = _spoil(workloads.c5_batch(n_pairs=900, lo=20, hi=420), rng)
    monkeypatch.setenv("ALN_CHUNK_CELLS", str(b.cells // 23))
    ref = _check_batch(orc, b, _ffi.CORE_LOCAL, 11, 2, blosum62)
    visible = list(range(torch.cuda.device_count()))
    for devs in (visible, [0, 0, 0]):
        got = align_batch(b, _ffi.CORE_LOCAL, 11, 2, blosum62, devices=devs)
        assert (got.results == ref.results).all(), devs
        for i in range(len(b)):
            assert all((x == y).all() for x, y in zip(got.aligned(i), ref.aligned(i))), (devs, i)
    cap = 2 * (b.q_len + b.t_len + np.uint64(2)) + np.uint64(3)
    off = np.zeros(len(b), dtype=np.uint64)
    off[::-1][1:] = np.cumsum(cap[::-1])[:-1]
    got = align_batch(b, _ffi.CORE_LOCAL, 11, 2, blosum62, devices=[0, 0, 0], tb_off=off)
    assert (got.results == ref.results).all()
    for i in range(0, len(b), 3):
        assert all((x == y).all() for x, y in zip(got.aligned(i), ref.aligned(i))), i
    # single calls on a multi-device context take the devices in turn
    lib = _ffi.load()
    assert lib.aln_device_count(runtime.context_multi([0, 0, 0])) == 3
    q, t = b.query(1), b.target(1)
    for _ in range(4):
        res, qa, ta, _, _ = runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, blosum62, device=None)
    want = orc.align(orc.CORE_LOCAL, q, t, 11, 2, blosum62)
    assert res.score == want["score"] and qa.tolist() == want["qa"].tolist()


def test_concurrent_callers_share_the_slot_pool(orc, blosum62):
    """Ten host threads call the library at once (statistics/mod.rs:255-286 runs ten aligner threads): the context's slots are
    leased per call, results equal the serial ones."""
    import threading
    rng = np.random.default_rng(5)
    batches = [workloads.c5_batch(n_pairs=40 + 3 * j, lo=20 + j, hi=300) for j in range(10)]
    pairs = [(rng.integers(0, 20, 200 + 10 * j).astype(np.uint8), rng.integers(0, 20, 150 + 7 * j).astype(np.uint8)) for j in range(10)]
    serial_b = [align_batch(b, _ffi.CORE_LOCAL, 11, 2, blosum62) for b in batches]
    serial_p = [runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, blosum62) for q, t in pairs]
    out_b, out_p, errs = [None] * 10, [None] * 10, []

    def work(j):
        try:
            for _ in range(3):
                out_b[j] = align_batch(batches[j], _ffi.CORE_LOCAL, 11, 2, blosum62)
                out_p[j] = runtime.align_pair(_ffi.CORE_LOCAL, pairs[j][0], pairs[j][1], 11, 2, blosum62)
        except Exception as e:            # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(j,)) for j in range(10)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for j in range(10):
        assert (out_b[j].results == serial_b[j].results).all(), j
        for i in range(len(batches[j])):          # (bytes of tb past aln_len are unspecified)
            assert all((x == y).all() for x, y in zip(out_b[j].aligned(i), serial_b[j].aligned(i))), (j, i)
        a, b_ = out_p[j], serial_p[j]
        assert (a[0].score, a[0].end_y, a[0].end_x, a[0].aln_len) == (b_[0].score, b_[0].end_y, b_[0].end_x, b_[0].aln_len)
        assert a[1].tolist() == b_[1].tolist() and a[2].tolist() == b_[2].tolist()
    q, t = pairs[0]
    want = orc.align(orc.CORE_LOCAL, q, t, 11, 2, blosum62)
    assert serial_p[0][0].score == want["score"] and serial_p[0][1].tolist() == want["qa"].tolist()


def test_pwm_windows_by_offsets_equal_copied_windows(orc):
    """engine/calc.rs:111-124 walks a chromosome in overlapping windows: align_window_offsets takes (start, length) into the one
    array -- same results as the windows copied out one by one, and as the oracle."""
    from aligner_amd.pwm import align_window_offsets, align_windows
    rng = np.random.default_rng(77)
    pwm = rng.integers(-2, 3, (4, 40)).astype(np.float64)
    chrom = rng.integers(0, 4, 9000).astype(np.uint8)
    starts = np.arange(0, 8000, 13, dtype=np.uint64)
    lens = rng.integers(1, 120, len(starts)).astype(np.uint64)
    res, alns = align_window_offsets(chrom, starts, lens, 3, 1, pwm)
    res2, alns2 = align_windows([chrom[int(a):int(a + b)] for a, b in zip(starts, lens)], 3, 1, pwm)
    assert (res == res2).all()
    for i in range(0, len(starts), 5):
        ref = orc.align_pwm(chrom[int(starts[i]):int(starts[i] + lens[i])], 3, 1, pwm)
        assert res["f"][i] == ref["f"] and alns[i].coords == ref["coords"] == alns2[i].coords
        assert alns[i].numbered.tolist() == ref["numbered"].tolist() == alns2[i].numbered.tolist()
        assert alns[i].query.tolist() == ref["qal"].tolist()


def test_c3_full_batch_matches_oracle(orc):
    """BASELINE C3 at full size (10 000 read pairs of 150 bp, core global, +5/-4, 10/1): every summary and both strings
    against the oracle run on the same batch (2.3e8 cells: seconds on the host)."""
    _check_batch(orc, workloads.c3_batch(10000), _ffi.CORE_GLOBAL, 10, 1, nucleotide_matrix())
