"""aligner-cli on the GPU path: FASTA (exactly two records) -> alignment midline.

Mirror of aligner-core/src/bin/aligner-cli/{main,args}.rs: flags -i/--input, -d/--deletions (11), -e/--extension (2),
-g/--global, -o/--output (parsed, never written -- as in the reference); Protein alphabet, BLOSUM62 as embedded; prints
the `{:?}` of `result.alignment.get_alignment(blosum62)` (main.rs:53), i.e. the Rust Debug form of a Vec<Protein>.
"""
import argparse
import sys

from .enums import ANY, BLANK, POS, Protein
from .fasta import read_fasta
from .matrices import get_blosum62
from .simple import SimpleGlobalAligner, SimpleLocalAligner


def debug_vec(codes):
    """`format!("{:?}", Vec<Protein>)`: variant names, e.g. [A, Blank, W, Pos]."""
    names = []
    for c in codes:
        c = int(c)
        names.append("Blank" if c == BLANK else "Pos" if c == POS else "Any" if c >= ANY or c >= 24 else Protein.letters[c])
    return "[" + ", ".join(names) + "]"


def main(argv=None):
    ap = argparse.ArgumentParser(prog="aligner-cli")
    ap.add_argument("-i", "--input", required=True)
    ap.add_argument("-d", "--deletions", type=float, default=11.0)
    ap.add_argument("-e", "--extension", type=float, default=2.0)
    ap.add_argument("-g", "--global", dest="global_", action="store_true")
    ap.add_argument("-o", "--output", default="out/result.txt")
    args = ap.parse_args(argv)
    seqs = read_fasta(args.input)
    if len(seqs) != 2:
        raise SystemExit("There's should be 2 sequences, not %d" % len(seqs))      # main.rs:31-33
    blosum62 = get_blosum62()
    query, target = seqs[0].seq.decode("utf-8"), seqs[1].seq.decode("utf-8")
    cls = SimpleGlobalAligner if args.global_ else SimpleLocalAligner
    result = cls.from_str_seqs(query, target).perform_alignment(args.deletions, args.extension, blosum62, None)
    print(debug_vec(result.alignment.get_alignment(blosum62)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
