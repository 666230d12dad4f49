"""Synthetic inputs of the benchmark configurations.

PRNG = splitmix64, residue = next() % A, seeds fixed per configuration and echoed by the harness.  "Homolog"
variants derive the target from the query with substitutions and indels so that a long real alignment exists.
"""
import numpy as np

from .batch import PairBatch

SEED_C2 = 0xA11C0002
SEED_C3 = 0xA11C0003
SEED_C4 = 0xA11C0004
SEED_C5 = 0xA11C0005
_MASK = (1 << 64) - 1


def splitmix64(seed, n, start=0):
    """Outputs start .. start+n-1 of splitmix64 seeded with `seed` (vectorised: state_i = seed + (i+1)*gamma).
    `n` may also be an integer index array (arbitrary output positions)."""
    with np.errstate(over="ignore"):
        gamma = np.uint64(0x9E3779B97F4A7C15)
        idx = np.arange(1, n + 1, dtype=np.uint64) + np.uint64(start) if np.isscalar(n) else \
            np.asarray(n).astype(np.uint64) + np.uint64(1)
        z = np.uint64(seed & _MASK) + gamma * idx
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _synth_ranges(seed, alphabet_size, src_off, dst_off, lens, out):
    """out[dst_off[j] + k] = splitmix64(seed) output (src_off[j] + k) % A -- native helper (csrc/aln_synth.c)."""
    import ctypes as C
    import os
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libaln_synth.so")
    if not os.path.exists(lib):
        # built by aligner_amd/build.py (__graft_entry__.build()) before anything touches the GPU -- never compiled here:
        # under torch.distributed.run N ranks would race on the same output file
        raise ImportError("aligner_amd: %s is missing -- run `python -m aligner_amd.build`" % lib)
    src_off = np.ascontiguousarray(src_off, dtype=np.int64)
    dst_off = np.ascontiguousarray(dst_off, dtype=np.int64)
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    C.CDLL(lib).aln_synth_ranges(C.c_uint64(seed & _MASK), C.c_uint32(alphabet_size), C.c_void_p(src_off.ctypes.data),
                                 C.c_void_p(dst_off.ctypes.data), C.c_void_p(lens.ctypes.data),
                                 C.c_size_t(len(lens)), C.c_void_p(out.ctypes.data))


def random_codes(seed, n, alphabet_size):
    return (splitmix64(seed, n) % np.uint64(alphabet_size)).astype(np.uint8)


def mutate(codes, seed, alphabet_size, sub_rate, indel_rate, out_len=None):
    """Target = query with substitutions and indels (deterministic); trimmed / padded to out_len if given."""
    n = len(codes)
    r = splitmix64(seed, 3 * n + 8)
    u = (r[:n] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    newc = (r[n:2 * n] % np.uint64(alphabet_size)).astype(np.uint8)
    u2 = (r[2 * n:3 * n] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    out = []
    for i in range(n):
        if u[i] < indel_rate / 2:
            continue                                # deletion
        c = newc[i] if u[i] < indel_rate / 2 + sub_rate else codes[i]
        out.append(c)
        if u2[i] < indel_rate / 2:
            out.append(newc[(i * 7 + 3) % n])       # insertion
    out = np.array(out, dtype=np.uint8)
    if out_len is not None:
        if len(out) >= out_len:
            out = out[:out_len]
        else:
            pad = random_codes(seed ^ 0x5EED, out_len - len(out), alphabet_size)
            out = np.concatenate([out, pad])
    return out


def c2_pair(homolog=False, n=1000):
    """C2: one 1k x 1k protein pair (20-letter uniform); homolog = 10 % substitutions + 2 % indels."""
    q = random_codes(SEED_C2, n, 20)
    t = mutate(q, SEED_C2 ^ 0x77, 20, 0.10, 0.02) if homolog else random_codes(SEED_C2 ^ 0xFFFF, n, 20)
    return q, t


def c3_batch(n_pairs=10000, length=150):
    """C3: n_pairs nucleotide read pairs of 150 bp, mate = read with 5 % substitutions + 1 % indels."""
    reads = random_codes(SEED_C3, n_pairs * length, 4).reshape(n_pairs, length)
    pairs = []
    for i in range(n_pairs):
        mate = mutate(reads[i], SEED_C3 + 1 + i, 4, 0.05, 0.01, out_len=length)
        pairs.append((reads[i], mate))
    return PairBatch.from_pairs(pairs)


def c4_pair(homolog=False, n=10000):
    """C4: one 10k x 10k protein pair."""
    q = random_codes(SEED_C4, n, 20)
    t = mutate(q, SEED_C4 ^ 0x77, 20, 0.10, 0.02) if homolog else random_codes(SEED_C4 ^ 0xFFFF, n, 20)
    return q, t


def c5_lengths(n_pairs=100000, lo=200, hi=2000):
    r = splitmix64(SEED_C5, 2 * n_pairs)
    span = np.uint64(hi - lo + 1)
    qlen = (r[:n_pairs] % span).astype(np.int64) + lo
    tlen = (r[n_pairs:] % span).astype(np.int64) + lo
    return qlen, tlen


def c5_batch(n_pairs=100000, lo=200, hi=2000, indices=None):
    """C5: protein pairs with both lengths iid uniform in [lo, hi].  `indices` selects a shard of the global batch;
    residue g of the global packed buffer is splitmix64(SEED_C5 + 1) output g, so a pair's residues do not depend
    on which shard it lands in."""
    qlen, tlen = c5_lengths(n_pairs, lo, hi)
    gtot = qlen + tlen
    goff = np.zeros(n_pairs + 1, dtype=np.int64)
    goff[1:] = np.cumsum(gtot)
    idx = np.arange(n_pairs) if indices is None else np.asarray(indices, dtype=np.int64)
    ql, tl = qlen[idx], tlen[idx]
    tot = ql + tl
    off = np.zeros(len(idx) + 1, dtype=np.int64)
    off[1:] = np.cumsum(tot)
    seqs = np.empty(int(off[-1]), dtype=np.uint8)
    _synth_ranges(SEED_C5 + 1, 20, goff[idx], off[:-1], tot, seqs)
    q_off = off[:-1]
    t_off = off[:-1] + ql
    return PairBatch(seqs, q_off, ql, t_off, tl)
