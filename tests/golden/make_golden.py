"""Regenerates tests/golden/legacy_kat.json from the reference's own known-answer tests.

Reads /root/reference/src/tests/test_alignment.rs AS TEXT (the reference is Rust and cannot be built or
imported here) and transcribes the literal expected matrices / alignments it holds (test_alignment.rs:14-67
global, :106-159 local) into a JSON fixture.  Only data is extracted -- numbers, Direction names, Protein
names -- no reference source text is kept.  Inputs come from examples/book_example_1.fasta (HEAGAWGHEE / PAWHEAE).

BLOSUM50 itself is not in the reference tree (load_blosum50 belongs to a missing module); the 6x6 sub-table
over {A,E,G,H,P,W} below is the standard NCBI BLOSUM50 restricted to the residues of this example.  It is
pinned by the golden H matrices: any wrong entry makes tests/test_oracle_golden.py fail.

Run in the dev container only (needs /root/reference): python tests/golden/make_golden.py
"""
import json
import os
import re

REF = "/root/reference/src/tests/test_alignment.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "legacy_kat.json")

DIR = {"Top": 0, "Left": 1, "Diagonal": 2, "Beginning": 3}
# aligner-core/src/enums.rs:56-84 codes
PROT = {c: i for i, c in enumerate("ARNDCQEGHILKMFPSTWYVBJZX")}
PROT["Blank"] = 98

# standard NCBI BLOSUM50, residues A E G H P W only (symmetric)
B50_SUB = {
    ("A", "A"): 5, ("A", "E"): -1, ("A", "G"): 0, ("A", "H"): -2, ("A", "P"): -1, ("A", "W"): -3,
    ("E", "E"): 6, ("E", "G"): -3, ("E", "H"): 0, ("E", "P"): -1, ("E", "W"): -3,
    ("G", "G"): 8, ("G", "H"): -2, ("G", "P"): -2, ("G", "W"): -3,
    ("H", "H"): 10, ("H", "P"): -2, ("H", "W"): -3,
    ("P", "P"): 10, ("P", "W"): -4,
    ("W", "W"): 15,
}


def section(text, start_marker, end_marker):
    a = text.index(start_marker)
    b = text.index(end_marker, a)
    return text[a:b]


def parse_case(body):
    am = section(body, "alignment_matrix: array![", "direction_matrix:")
    rows = re.findall(r"\[([-\d,\s]+)\]", am)
    H = [[int(v) for v in r.replace("\n", " ").split(",") if v.strip()] for r in rows]
    dm = section(body, "direction_matrix: array![", "optimal_alignment:")
    toks = re.findall(r"Top|Left|Diagonal|Beginning", dm)
    w = len(H[0])
    D = [[DIR[t] for t in toks[i * w:(i + 1) * w]] for i in range(len(H))]
    assert len(toks) == w * len(H)
    oa = section(body, "optimal_alignment: (", "});")
    vecs = re.findall(r"vec!\[(.*?)\]", oa, flags=re.S)
    al = [[PROT[n] for n in re.findall(r"Protein::(\w+)", v)] for v in vecs]
    return H, D, al


def main():
    text = open(REF).read()
    g = section(text, "fn test_global_alignment", "fn test_local_alignment")
    l = text[text.index("fn test_local_alignment"):]
    gH, gD, gA = parse_case(g)
    lH, lD, lA = parse_case(l)
    fasta = open("/root/reference/examples/book_example_1.fasta").read().split(">")[1:]
    seqs = ["".join(r.splitlines()[1:]) for r in fasta]
    query = seqs[0]
    target = "".join(c for c in seqs[1] if c in PROT)
    m = [[0] * 24 for _ in range(24)]
    for (a, b), v in B50_SUB.items():
        m[PROT[a]][PROT[b]] = v
        m[PROT[b]][PROT[a]] = v
    out = {
        "source": "src/tests/test_alignment.rs:14-67 (global), :106-159 (local); examples/book_example_1.fasta",
        "query": query, "target": target, "gap": 8,
        "blosum50_sub": m,
        "global": {"H": gH, "D": gD, "query_aligned": gA[0], "target_aligned": gA[1]},
        "local": {"H": lH, "D": lD, "query_aligned": lA[0], "target_aligned": lA[1]},
    }
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", OUT, "query", query, "target", target)


if __name__ == "__main__":
    main()
