/* aln_synth.c -- synthetic residue generator for the benchmark inputs.
 * splitmix64, residue = next() % A.  Host-side input generation only; not part of the DP path. */
#include <stddef.h>
#include <stdint.h>

static inline uint64_t splitmix_at(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* out[dst_off[j] + k] = splitmix64(seed) output (src_off[j] + k) % A   for k < len[j], j < n_ranges */
void aln_synth_ranges(uint64_t seed, uint32_t A, const int64_t *src_off, const int64_t *dst_off, const int64_t *len,
                      size_t n_ranges, uint8_t *out)
{
    for (size_t j = 0; j < n_ranges; ++j) {
        uint8_t *o = out + dst_off[j];
        const uint64_t s = (uint64_t)src_off[j];
        for (int64_t k = 0; k < len[j]; ++k) o[k] = (uint8_t)(splitmix_at(seed, s + (uint64_t)k) % A);
    }
}
