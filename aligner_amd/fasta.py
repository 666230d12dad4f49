"""FASTA ingest for the DP path: records -> residue code arrays.

The reference reads FASTA with seq_io 0.3.1 (Cargo.lock:1897; not vendored) at aligner-core/src/bin/aligner-cli/main.rs:21-37
and encodes with BioData::str_to_vec (enums.rs:266-277 / :428-439).  Restated behaviour of `seq_io::fasta::Reader::records()`:
a record starts at a line beginning with '>', the rest of that line (minus the line terminator) is the head, every
following line up to the next '>' line is sequence; `record.seq` is the sequence with line terminators ('\n', '\r\n')
removed and NOTHING else -- spaces or '-' inside a line stay and make str_to_vec fail (CharIsNotMatchable), exactly as
examples/human_gene_example.fasta (a stray space on one line) does in the reference.
A file that does not start with '>' is an error.
"""
import numpy as np

from .batch import PairBatch
from .enums import Protein
from .errors import AlignerError, ErrorKind


class FastaRecord:
    def __init__(self, head, seq):
        self.head = head
        self.seq = seq          # bytes, line terminators removed

    @property
    def id(self):
        return self.head.split(b" ", 1)[0]


def parse_fasta(data):
    """bytes/str -> [FastaRecord]."""
    if isinstance(data, str):
        data = data.encode("utf-8")
    records = []
    head, chunks, started = None, [], False
    for line in data.split(b"\n"):
        if line.endswith(b"\r"):
            line = line[:-1]
        if line.startswith(b">"):
            if head is not None:
                records.append(FastaRecord(head, b"".join(chunks)))
            head, chunks, started = line[1:], [], True
        elif not started:
            if line.strip() == b"":
                continue            # leading empty lines are skipped by seq_io
            raise ValueError("FASTA parse error: expected '>' at record start")
        else:
            chunks.append(line)
    if head is not None:
        records.append(FastaRecord(head, b"".join(chunks)))
    return records


def read_fasta(path):
    with open(path, "rb") as f:
        return parse_fasta(f.read())


def encode_records(records, alphabet=Protein, strict=True):
    """Residue codes of every record.  strict=True is `str_to_vec` (any unknown byte -> CharIsNotMatchable);
    strict=False is `from_u8_vec` (Protein still errors, DNA silently skips: enums.rs:292-303 / :454-467)."""
    out = []
    for r in records:
        if strict:
            try:
                text = r.seq.decode("utf-8")
            except UnicodeDecodeError:
                raise AlignerError(ErrorKind.CharIsNotMatchable)
            out.append(alphabet.str_to_vec(text))
        else:
            out.append(alphabet.from_u8_vec(r.seq))
    return out


def pairs_from_fasta(path, alphabet=Protein, strict=True):
    """Consecutive records (1,2), (3,4), ... as a PairBatch -- the ingest step in front of the batch driver."""
    codes = encode_records(read_fasta(path), alphabet, strict)
    if len(codes) % 2:
        raise ValueError("odd number of FASTA records")
    return PairBatch.from_pairs(zip(codes[0::2], codes[1::2]))
