"""Host-side logic that needs no GPU: alphabets, error surface, result post-processing, batch packing,
the direction-region layout arithmetic, and that the C-ABI library loads and exports what the header declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from aligner_amd import _ffi, workloads
from aligner_amd.alignment import Alignment
from aligner_amd.batch import RESULT_DTYPE, PairBatch
from aligner_amd.enums import ANY, BLANK, DNA, POS, Direction, Protein
from aligner_amd.errors import AlignerError, ErrorKind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_direction_discriminants():
    # enums.rs:9-15 -- also the 2-bit code the kernels store
    assert [int(d) for d in (Direction.Top, Direction.Left, Direction.Diagonal, Direction.Beginning)] == [0, 1, 2, 3]


def test_protein_codec():
    # enums.rs:56-84, :201-264
    assert Protein.str_to_vec("ARNDCQEGHILKMFPSTWYVBJZX").tolist() == list(range(24))
    assert Protein.str_to_vec("_+").tolist() == [BLANK, POS]
    assert Protein.vec_to_str([0, 23, BLANK, POS, ANY, 57]) == "AX_+**"
    assert Protein.volume() == 24 and Protein.blank() == 98 and Protein.pos() == 99
    for bad in ("a", "-", " ", "O", "U", "*", "é"):
        with pytest.raises(AlignerError) as e:
            Protein.str_to_vec("AC" + bad)
        assert e.value.kind == ErrorKind.CharIsNotMatchable
    with pytest.raises(AlignerError):
        Protein.from_u8_vec(b"AC-")          # enums.rs:292-303 errors
    assert Protein.match_with_char("W") == 17 and Protein.convert_to_char(17) == "W"


def test_dna_codec():
    # enums.rs:139-147 A,T,C,G order; :454-467 from_u8_vec silently skips unknown bytes
    assert DNA.str_to_vec("ATCG").tolist() == [0, 1, 2, 3]
    assert DNA.volume() == 4
    with pytest.raises(AlignerError):
        DNA.str_to_vec("ATN")
    assert DNA.from_u8_vec(b"AT NCG\n").tolist() == [0, 1, 2, 3]
    codes, freqs = DNA.from_u8_vec_with_freqs(b"AATT-C")
    assert codes.tolist() == [0, 0, 1, 1, 2] and np.allclose(freqs, [0.4, 0.4, 0.2, 0.0])


def test_blosum62_as_embedded(blosum62):
    """lib.rs:61-90 indexed by enum code: J/Z/X read NCBI's Z/X/* rows."""
    c = Protein.match_with_char
    assert blosum62.shape == (24, 24)
    assert blosum62[c("W"), c("W")] == 11 and blosum62[c("A"), c("R")] == -1
    assert blosum62[c("X"), c("X")] == 1 and blosum62[c("X"), c("A")] == -4       # the '*' row
    assert blosum62[c("Z"), c("Z")] == -1                                          # NCBI's X row
    assert blosum62[c("J"), c("J")] == 4 and blosum62[c("J"), c("E")] == 4         # NCBI's Z row


def test_midline_and_frequency_matrix(orc, blosum62):
    # alignment.rs:13-43, checked against the oracle's restatement
    qa = Protein.str_to_vec("HEAGAWGHE_EE")
    ta = Protein.str_to_vec("P_A__W_HEAEE")
    a = Alignment(Protein, qa, ta, ((1, 10), (1, 7)), 0.0)
    assert a.midline_str(blosum62) == "__A__W_HE_EE"
    assert (a.get_alignment(blosum62) == orc.midline(qa, ta, blosum62)).all()
    assert (a.get_frequency_matrix() == orc.frequency_matrix(qa, ta, 24)).all()
    qa2, ta2 = Protein.str_to_vec("KR_W"), Protein.str_to_vec("RKAW")
    assert Alignment(Protein, qa2, ta2, None, 0).midline_str(blosum62) == "++_W"   # K-R scores +2 -> Pos


def test_pair_batch_packing():
    rng = np.random.default_rng(0)
    pairs = [(rng.integers(0, 20, n).astype(np.uint8), rng.integers(0, 20, m).astype(np.uint8))
             for n, m in ((3, 5), (0, 2), (7, 1))]
    b = PairBatch.from_pairs(pairs)
    assert len(b) == 3 and b.cells == 15 + 0 + 7
    for i, (q, t) in enumerate(pairs):
        assert (b.query(i) == q).all() and (b.target(i) == t).all()
    off, total = b.tb_layout()
    assert off.tolist() == [0, 20, 28] and total == 20 + 8 + 20
    sub = b.select([2, 0])
    assert (sub.query(0) == pairs[2][0]).all() and (sub.target(1) == pairs[0][1]).all()


def test_workload_generators_are_deterministic():
    assert workloads.splitmix64(0, 3).tolist() == [16294208416658607535, 7960286522194355700, 487617019471545679]
    a, b = workloads.c5_batch(n_pairs=50), workloads.c5_batch(n_pairs=50)
    assert (a.seqs == b.seqs).all() and a.q_len.min() >= 200 and a.t_len.max() <= 2000
    shard = workloads.c5_batch(n_pairs=50, indices=[7, 3])
    assert (shard.query(0) == a.query(7)).all() and (shard.target(1) == a.target(3)).all()
    ql, tl = workloads.c5_lengths()
    assert abs(float((ql * tl).sum()) - 1.21e11) < 0.02e11            # E[cells] ~ 1.21e11
    q, t = workloads.c2_pair(homolog=True)
    assert len(q) == 1000 and 900 < len(t) < 1100
    c3 = workloads.c3_batch(20)
    assert set(c3.q_len.tolist()) == {150} and set(c3.t_len.tolist()) == {150} and c3.seqs.max() <= 3


def test_result_record_layout_matches_header():
    hdr = open(os.path.join(ROOT, "include", "aligner_hip.h")).read()
    body = hdr[hdr.index("typedef struct aln_pair_result {"):hdr.index("} aln_pair_result;")]
    names = re.findall(r"(\w+)\s*(?:,|;)", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert names == [n for n in RESULT_DTYPE.names]
    assert [f for f, _ in _ffi.PairResult._fields_] == list(RESULT_DTYPE.names)
    assert C.sizeof(_ffi.PairResult) == RESULT_DTYPE.itemsize == 48


def test_native_library_loads_and_exports_every_declared_symbol():
    """No compute calls here (no GPU): the library must load and export exactly what include/aligner_hip.h declares."""
    from aligner_amd import build as native_build
    native_build.build()
    lib = _ffi.load()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "aligner_hip.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(aln_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(_ffi.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.aln_abi_version() == 2


def test_header_compiles_as_c99_and_links():
    """tests/abi_harness.c = include/aligner_hip.h used from C: built with -std=c99 -Wall -Werror (record sizes and offsets are pinned at
    compile time there), linked against the library; without a case file it only loads the library and compares ABI versions."""
    import subprocess
    from aligner_amd import build as native_build
    native_build.build()
    exe = native_build.build_harness()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_version_header 2 sizeof_result 48 sizeof_params %d" % C.sizeof(_ffi.Params) in out.stdout
    assert "abi_version_library 2" in out.stdout


def test_no_gpu_means_loud_failure_not_fallback(blosum62):
    """Without a device the product path must raise; it must never produce an answer some other way."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from aligner_amd.errors import DeviceError
    from aligner_amd.simple import SimpleLocalAligner
    with pytest.raises(DeviceError):
        SimpleLocalAligner.from_str_seqs("HEAGAWGHEE", "PAWHEAE").perform_alignment(11.0, 2.0, blosum62)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "aligner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
                assert "aligner_oracle" not in src, f


def test_transform_matrix_properties(blosum62):
    """aligner-helpers/src/matrices/mod.rs:19-68: the transformed matrix has expectation k_d under p and norm^2 r^2."""
    from aligner_amd.heuristic import find_roots_quadratic, get_threshold, transform_matrix
    freqs = np.full(24, 1 / 24.0)
    kd, r2 = -0.5, 576.0
    m = transform_matrix(blosum62, kd, r2, freqs)
    p = np.outer(freqs, np.full(24, 1 / 24.0))
    assert abs((p * m).sum() - kd) < 1e-9 and abs((m * m).sum() - r2) < 1e-6
    assert find_roots_quadratic(1.0, -3.0, 2.0) == (1.0, 2.0) and find_roots_quadratic(1.0, 2.0, 1.0) == (-1.0,)
    assert find_roots_quadratic(1.0, 0.0, 1.0) == () and get_threshold(24) == 24.6 and get_threshold(7) == 0.0
    with pytest.raises(Exception) as e:        # |k_d| > sqrt(r^2 * sum p^2): no real root -> Err(WrongMatrixSpecified)
        transform_matrix(blosum62, -1.5, r2, freqs)
    assert type(e.value).__name__ == "WrongMatrixSpecified"


def test_generated_asm_is_in_sync_with_its_generator(tmp_path):
    """aligner_amd/csrc/aln_single_unit.inc is generated by tools/gen_single_asm.py and committed: regenerate and compare;
    and check the invariants the kernel relies on: every in-loop LDS wait is an explicit count, the loop drains before it
    ends, and each variant names its poll loops uniquely."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_single_asm", os.path.join(ROOT, "tools", "gen_single_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(ROOT, "aligner_amd", "csrc", "aln_single_unit.inc")).read()
    for sem, prefix in (("LOCAL", ""), ("GLOBAL", "G")):
      for R in (1, 2):
        for kind in ("FIRST", "MID", "LAST"):
            for masked in ((False,) if kind == "FIRST" else (False, True)):
                lines = gen.loop(R, kind, masked, sem)
                assert "WAIT" not in lines and not any(";M" in ln for ln in lines)
                assert lines[-1] == "s_waitcnt vmcnt(0) lgkmcnt(0)"
                labels = [ln for ln in lines if ln.endswith(":")]
                assert len(labels) == len(set(labels))
                name = "ALN_%s%s_ASM_R%d_%s" % (prefix, "MASKED" if masked else "STEADY", R, kind)
                body = committed[committed.index("#define " + name):]
                body = body[:body.index("\n\n")]
                got = [ln.strip().rstrip("\\").strip().strip('"').replace("\\n\\t", "") for ln in body.splitlines()[1:]]
                assert got == lines, name


def test_c5_digest_fixture_is_the_oracles():
    """tests/golden/c5_100k_digest.npz (what the GPU suite checks the full C5 batch against) really is the CPU oracle's
    output: one block of 1000 pairs is regenerated here (make_c5_golden.py made all 100)."""
    import importlib.util
    import os
    import oracle as orc
    from aligner_amd import workloads
    from aligner_amd.matrices import get_blosum62
    here = os.path.dirname(__file__)
    spec = importlib.util.spec_from_file_location("make_c5_golden", os.path.join(here, "golden", "make_c5_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    g = np.load(os.path.join(here, "golden", "c5_100k_digest.npz"))
    k = 57
    b = workloads.c5_batch(100000, indices=np.arange(k * 1000, (k + 1) * 1000))
    ref, tb, tb_off = orc.align_batch(orc.CORE_LOCAL, b.seqs, b.q_off, b.q_len, b.t_off, b.t_len, 11, 2, get_blosum62(), n_threads=8)
    summ = np.zeros((1000, 6), dtype=np.int32)
    strings = []
    for i in range(1000):
        r = ref[i]
        summ[i] = (int(r.score), r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len)
        cap = int(b.q_len[i] + b.t_len[i]) + 2
        o = int(tb_off[i])
        strings.append((tb[o:o + r.aln_len], tb[o + cap:o + cap + r.aln_len]))
    assert mk.block_digest(summ, strings) == g["digests"][k].tobytes()
    assert (summ[:, 0] == g["scores"][k * 1000:(k + 1) * 1000]).all()


def test_evd_fit_keeps_the_references_loop_scoped_rebinding():
    """statistics/mod.rs:69 re-binds (k, lambda) INSIDE the loop body: every outer iteration restarts from the initial moment
    estimates, and the fall-through returns them.  The numpy restatement must agree with the scalar hand-run of the Rust
    logic (tests/pyref.py): on realistic score sets (one outer iteration: the log-likelihood is negative, so the relative
    test at :104 passes at once), on sets that take several outer iterations, and on sets that fall through the cap."""
    import pyref
    from aligner_amd import statistics as st
    cap = 30
    seen = set()
    old = st.MAXITER
    st.MAXITER = cap                          # same cap for the outer and the inner loops on both sides
    try:
        for case in range(16):
            rng = np.random.default_rng(3 if case >= 4 else 100 + case)
            if case < 4:                      # Gumbel-ish scores like the shuffled alignments produce
                n, ql = 60 + 10 * case, 150 + 20 * case
                lengths = rng.integers(ql - 6, ql + 1, size=n)
                scores = np.round(rng.gumbel(30.0 + 5 * case, 4.0 + case, size=n))
            else:                             # tiny-variance scores around zero: positive log-likelihood, several iterations
                for c in range(case - 3):
                    n, ql = 40 + 5 * c, 100 + 10 * c
                    lengths = rng.integers(ql - 6, ql + 1, size=n)
                    scores = np.round(rng.normal(0.0, [0.05, 0.1, 0.2, 0.3][c % 4], size=n), 3)
            k, lam, h, iters = pyref.evd_params(ql, lengths.tolist(), scores.tolist(), maxiter=cap)
            got = st.calculate_distribution_params(ql, lengths, scores)
            seen.add("one" if iters == 1 else "cap" if iters > cap else "several")
            for a, b in ((got.k, k), (got.lambda_, lam), (got.h, h)):
                assert (np.isnan(a) and np.isnan(b)) or a == pytest.approx(b, rel=1e-9, abs=0), (case, iters, got.k, k, got.lambda_, lam, got.h, h)
    finally:
        st.MAXITER = old
    assert seen == {"one", "several", "cap"}, seen


def test_in_library_chunking_for_any_device_count():
    """aln_plan_chunks (host arithmetic, no GPU): the chunks a batch call is cut into cover the pairs exactly once, in the
    caller's order, stay within the cell bounds, and give every device of an 8-GPU context several chunks to take."""
    lib = _ffi.load()
    qlen, tlen = workloads.c5_lengths(100000)
    ql, tl = qlen.astype(np.uint64), tlen.astype(np.uint64)
    cells = (qlen * tlen).astype(np.float64)
    p = _ffi.Params(_ffi.CORE_LOCAL, 0, 11.0, 2.0, None, 24, 24, 24, 3, 98, 0, 0, 0, 0)
    for ndev in (1, 2, 3, 4, 8):
        cap = 4096
        first, count = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        n = lib.aln_plan_chunks(C.byref(p), ql.ctypes.data, tl.ctypes.data, len(ql), ndev, first.ctypes.data, count.ctypes.data, cap)
        first, count = first[:n].astype(np.int64), count[:n].astype(np.int64)
        assert first[0] == 0 and (first[1:] == np.cumsum(count)[:-1]).all() and count.sum() == len(ql)
        per = np.array([cells[f:f + c].sum() for f, c in zip(first, count)])
        assert per.max() <= 1.6e10 * 1.3 and per[:-1].min() >= 5e9 * 0.99
        assert n >= 3 * ndev or per.max() <= 5.1e9          # every device gets about three chunks, or the chunks are at the floor
        if ndev == 1:
            assert n == 8
    # a small batch is one chunk; nothing at all is no chunk
    assert lib.aln_plan_chunks(C.byref(p), ql.ctypes.data, tl.ctypes.data, 500, 8, None, None, 0) == 1
    assert lib.aln_plan_chunks(C.byref(p), ql.ctypes.data, tl.ctypes.data, 0, 1, None, None, 0) == 0
    # large pairs: a chunk holds two pairs per resident wave (6144) or the whole batch -- 1024 and 3072 pairs of 4200 x 4200 are one
    # chunk each (r02's bounds cut them into 4 and 11 chunks of 284 pairs: one wave per pair, nine strips one after the other);
    # 10 000 of them are cut at the 6.4e10-cell cap
    big = np.full(10000, 4200, np.uint64)
    assert lib.aln_plan_chunks(C.byref(p), big.ctypes.data, big.ctypes.data, 1024, 1, None, None, 0) == 1
    assert lib.aln_plan_chunks(C.byref(p), big.ctypes.data, big.ctypes.data, 3072, 1, None, None, 0) == 1
    first, count = np.zeros(64, np.uint64), np.zeros(64, np.uint64)
    n = lib.aln_plan_chunks(C.byref(p), big.ctypes.data, big.ctypes.data, 10000, 1, first.ctypes.data, count.ctypes.data, 64)
    assert n == 3 and int(count[:n].sum()) == 10000 and int(count[0]) * 4200 * 4200 <= 6.4e10 * 1.01
