// Probe: the 64-deep lane shift register (DPP wave_shl:1) + one coalesced 8-byte granule store per 64 steps, as FastStrip::flush_below
// does it.  Every column x in 1..N must end up as {value(x), tag}.  usage: ./shiftreg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long *row, unsigned N, unsigned tag)
{
    const int lane = threadIdx.x & 63;
    int outq = 0;
    const unsigned nsteps = ((N + 63 + 7) / 8) * 8;
    for (unsigned k = 0; k < nsteps; ++k) {
        if ((k & 63u) == 0 && k >= 64u) {
            const unsigned x = k - 126u + (unsigned)lane;
            if (x - 1u < N) __hip_atomic_store(row + x, ((unsigned long long)tag << 32) | (unsigned)outq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int bottom = (int)(1000u + k - (unsigned)lane + 1u);      // "value of column x = k - lane + 1" as computed by this lane
        outq = __builtin_amdgcn_update_dpp(bottom, outq, 0x130, 0xf, 0xf, false);
    }
    const unsigned kb_last = ((nsteps - 1u) / 64u) * 64u, beyond = kb_last >= 64u ? kb_last - 63u : 0u;
    const unsigned x = nsteps - 126u + (unsigned)lane;
    if (x - 1u < N && x > beyond) __hip_atomic_store(row + x, ((unsigned long long)tag << 32) | (unsigned)outq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main()
{
    const unsigned N = 261;
    unsigned long long *d;
    hipMalloc(&d, (N + 66) * 8);
    hipMemset(d, 0, (N + 66) * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, N, 0x80080082u);
    std::vector<unsigned long long> h(N + 66);
    hipMemcpy(h.data(), d, (N + 66) * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (unsigned x = 1; x <= N; ++x) {
        const unsigned tag = (unsigned)(h[x] >> 32), v = (unsigned)h[x];
        const unsigned want = 1000u + (x + 62u) - 63u + 1u;      // lane 63 at step x + 62
        if (tag != 0x80080082u || v != want) { if (bad < 20) printf("column %u: tag %#x value %u (want %u)\n", x, tag, v, want); ++bad; }
    }
    printf("N %u: %d bad columns\n", N, bad);
    return bad != 0;
}
