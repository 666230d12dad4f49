"""FASTA ingest + residue encoding + CLI formatting on the CPU.  The example files are the reference's own
inputs (examples/*.fasta), kept as data fixtures under tests/golden/."""
import os

import numpy as np
import pytest

from aligner_amd import cli
from aligner_amd.enums import DNA, Protein
from aligner_amd.errors import AlignerError, ErrorKind
from aligner_amd.fasta import encode_records, pairs_from_fasta, parse_fasta, read_fasta

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_parse_multiline_records():
    recs = parse_fasta(b">a desc\r\nHEAG\r\nAWGH\r\n>b\nPAW\nHEAE\n")
    assert [(r.id, r.seq) for r in recs] == [(b"a", b"HEAGAWGH"), (b"b", b"PAWHEAE")]
    assert parse_fasta(b"\n\n>x\nAC\n")[0].seq == b"AC"
    with pytest.raises(ValueError):
        parse_fasta(b"ACGT\n>x\nAC\n")


def test_reference_examples():
    prot = read_fasta(os.path.join(G, "protein.fasta"))
    assert [len(r.seq) for r in prot] == [340, 341] and prot[0].id.startswith(b"sp|A6NL46")
    q, t = encode_records(prot, Protein)
    assert Protein.vec_to_str(q[:10]) == "MRLCLIPWNT"
    # book_example_1.fasta: no trailing newline after the last record
    book = read_fasta(os.path.join(G, "book_example_1.fasta"))
    assert [r.seq for r in book] == [b"HEAGAWGHEE", b"PAWHEAE"]
    assert [Protein.vec_to_str(c) for c in encode_records(book, Protein)] == ["HEAGAWGHEE", "PAWHEAE"]
    with pytest.raises(AlignerError) as e:            # str_to_vec rejects anything outside the alphabet (enums.rs:266-277)
        encode_records(parse_fasta(b">x\nPAWHEAE---\n"), Protein)
    assert e.value.kind == ErrorKind.CharIsNotMatchable
    # human_gene_example.fasta is nucleotide with a stray space: strict encoding fails, from_u8_vec (DNA) skips it
    gene = read_fasta(os.path.join(G, "human_gene_example.fasta"))
    with pytest.raises(AlignerError):
        encode_records(gene, DNA, strict=True)
    lens = [len(c) for c in encode_records(gene, DNA, strict=False)]
    assert lens[0] == 1231 and lens[1] in (1020, 1021)
    with pytest.raises(AlignerError):
        encode_records(gene, Protein, strict=False)       # Protein::from_u8_vec errors (enums.rs:292-303)


def test_pairs_from_fasta_packs_consecutive_records():
    b = pairs_from_fasta(os.path.join(G, "protein.fasta"))
    assert len(b) == 1 and int(b.q_len[0]) == 340 and int(b.t_len[0]) == 341


def test_cli_debug_format():
    assert cli.debug_vec(Protein.str_to_vec("AW_+")) == "[A, W, Blank, Pos]"
