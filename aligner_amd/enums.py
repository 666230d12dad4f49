"""Data model of the hot path: Direction, Protein / DNA alphabets and the BioData codec.

Host-side mirror of aligner-core/src/enums.rs (Direction :9-53, Protein :55-136, DNA :138-179,
BioData :181-199 and its two impls :201-401 / :403-565).  Residues travel to the GPU as one byte
per residue holding the enum discriminant, exactly the `Into<usize>` value the reference indexes
its substitution matrix with.
"""
from enum import IntEnum

import numpy as np

from .errors import AlignerError, ErrorKind


class Direction(IntEnum):
    """enums.rs:9-15 -- the discriminants double as the 2-bit traceback code the kernels store."""
    Top = 0
    Left = 1
    Diagonal = 2
    Beginning = 3


BLANK = 98   # Protein::Blank / DNA::Blank (enums.rs:81,144)
POS = 99     # Protein::Pos / DNA::Pos     (enums.rs:82,145)
ANY = 100    # Protein::Any / DNA::Any (next discriminant after Pos)


class _Alphabet:
    """Shared BioData behaviour (enums.rs:181-199); subclasses fix the letters."""
    letters = ""
    name = ""
    # from_u8_vec: Protein errors on an unknown byte (enums.rs:292-303), DNA skips it (enums.rs:454-467)
    from_u8_skips_unknown = False

    @classmethod
    def _tables(cls):
        if "_enc" not in cls.__dict__:
            enc = np.full(256, 255, dtype=np.uint8)
            for i, ch in enumerate(cls.letters):
                enc[ord(ch)] = i
            enc[ord("_")] = BLANK
            enc[ord("+")] = POS
            dec = np.full(256, ord("*"), dtype=np.uint8)   # Any -> '*'
            for i, ch in enumerate(cls.letters):
                dec[i] = ord(ch)
            dec[BLANK] = ord("_")
            dec[POS] = ord("+")
            cls._enc, cls._dec = enc, dec
        return cls._enc, cls._dec

    @classmethod
    def volume(cls):
        return len(cls.letters)

    @classmethod
    def blank(cls):
        return BLANK

    @classmethod
    def pos(cls):
        return POS

    @classmethod
    def match_with_char(cls, symbol):
        enc, _ = cls._tables()
        o = ord(symbol)
        if o > 255 or enc[o] == 255:
            raise AlignerError(ErrorKind.CharIsNotMatchable)
        return int(enc[o])

    @classmethod
    def convert_to_char(cls, code):
        _, dec = cls._tables()
        return chr(dec[int(code) if int(code) < 256 else ANY])

    @classmethod
    def str_to_vec(cls, sequence):
        """enums.rs:266-277 / :428-439 -- any unknown char (lowercase, space, 'N', '-') is an error."""
        enc, _ = cls._tables()
        try:
            raw = np.frombuffer(sequence.encode("latin-1"), dtype=np.uint8)
        except UnicodeEncodeError:
            raise AlignerError(ErrorKind.CharIsNotMatchable)
        out = enc[raw]
        if (out == 255).any():
            raise AlignerError(ErrorKind.CharIsNotMatchable)
        return out.copy()

    @classmethod
    def vec_to_str(cls, codes):
        _, dec = cls._tables()
        c = np.asarray(codes, dtype=np.int64)
        c = np.where((c < 0) | (c > 255), ANY, c)
        return dec[c].tobytes().decode("latin-1")

    @classmethod
    def from_u8_vec(cls, raw):
        enc, _ = cls._tables()
        raw = np.frombuffer(bytes(raw), dtype=np.uint8)
        out = enc[raw]
        bad = out == 255
        if bad.any():
            if not cls.from_u8_skips_unknown:
                raise AlignerError(ErrorKind.CharIsNotMatchable)
            out = out[~bad]
        return out.copy()

    @classmethod
    def from_u8_vec_with_freqs(cls, raw):
        """enums.rs:305-323 / :469-487 -- unknown bytes are skipped, freqs = counts / kept length."""
        enc, _ = cls._tables()
        raw = np.frombuffer(bytes(raw), dtype=np.uint8)
        out = enc[raw]
        out = out[out != 255]
        freqs = np.zeros(cls.volume(), dtype=np.float64)
        # reference indexes freqs[v as usize]; '_' / '+' (98/99) would panic there -- same here
        np.add.at(freqs, out.astype(np.int64), 1.0)
        return out.copy(), freqs / float(len(out))

    @classmethod
    def random_seq(cls, length, rng=None):
        """enums.rs:365-374 -- uniform over 0..volume()."""
        rng = rng or np.random.default_rng()
        return rng.integers(0, cls.volume(), size=length, dtype=np.uint8)


class Protein(_Alphabet):
    letters = "ARNDCQEGHILKMFPSTWYVBJZX"   # codes 0..23, enums.rs:56-80
    name = "Protein"


class DNA(_Alphabet):
    letters = "ATCG"                       # codes 0..3 in A,T,C,G order, enums.rs:139-143
    name = "DNA"
    from_u8_skips_unknown = True
