// prio_probe.hip -- what does s_setprio do to VALU-bound waves that share a SIMD on gfx950?
// One workgroup = 4 waves = one wave per SIMD; 3 workgroups per CU, so every SIMD holds three waves, each running the same
// dependent-ish v_max3 / v_add loop (4 chains).  Workgroup j takes priority prio[(j / CUs) % 3] from the command line; every wave
// reports the shader time it needed for the same number of instructions.  Output: mean time per priority class.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/prio_probe tools/prio_probe.hip ; run: prio_probe 0 0 0 | 0 1 2 | 2 1 0 | 0 0 3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void probe(uint64_t *out, uint32_t iters, uint32_t cus, int p0, int p1, int p2, uint32_t chains)
{
    const uint32_t cls = (blockIdx.x / cus) % 3u;
    const int p = cls == 0 ? p0 : cls == 1 ? p1 : p2;
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 3) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
    int a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    const int s0 = (int)iters | 1, s1 = (int)blockIdx.x;
    __syncthreads();
    const uint64_t t0 = wall_clock64();
    if (chains == 9) {                          // rotation: the priority follows the 100 MHz clock, phase = class
        for (uint32_t i = 0; i < iters; i += 64) {
            const uint32_t slot = ((uint32_t)(wall_clock64() >> 13) + cls) % 3u;
            if (slot == 0) __builtin_amdgcn_s_setprio(2);
            else if (slot == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            for (uint32_t j = 0; j < 64; ++j)
                asm volatile("v_max3_i32 %0, %0, %4, %5\n\tv_max3_i32 %1, %1, %4, %5\n\tv_max3_i32 %2, %2, %4, %5\n\tv_max3_i32 %3, %3, %4, %5\n\t"
                             "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s0), "v"(s1));
        }
    } else if (chains == 1) {
        for (uint32_t i = 0; i < iters; ++i)
            asm volatile("v_max3_i32 %0, %0, %4, %5\n\tv_add_u32 %0, %0, %4\n\tv_max3_i32 %0, %0, %4, %5\n\tv_add_u32 %0, %0, %4\n\t"
                         "v_max3_i32 %0, %0, %4, %5\n\tv_add_u32 %0, %0, %4\n\tv_max3_i32 %0, %0, %4, %5\n\tv_add_u32 %0, %0, %4\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s0), "v"(s1));
    } else {
        for (uint32_t i = 0; i < iters; ++i)
            asm volatile("v_max3_i32 %0, %0, %4, %5\n\tv_max3_i32 %1, %1, %4, %5\n\tv_max3_i32 %2, %2, %4, %5\n\tv_max3_i32 %3, %3, %4, %5\n\t"
                         "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s0), "v"(s1));
    }
    const uint64_t t1 = wall_clock64();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63u) == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = t0; }
    if (a0 + a1 + a2 + a3 == 0x12345) out[0] = 0;
}

int main(int argc, char **argv)
{
    const int p0 = argc > 1 ? atoi(argv[1]) : 0, p1 = argc > 2 ? atoi(argv[2]) : 0, p2 = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t chains = argc > 4 ? (uint32_t)atoi(argv[4]) : 4u;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount, grid = cus * 3, iters = 200000;
    uint64_t *d;
    CHK(hipMalloc(&d, sizeof(uint64_t) * grid * 8));
    std::vector<uint64_t> h(grid * 8);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, 1000u, cus, p0, p1, p2, chains);
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, iters, cus, p0, p1, p2, chains);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h.data(), d, sizeof(uint64_t) * grid * 8, hipMemcpyDeviceToHost));
    uint64_t tmin = ~0ull;
    for (uint32_t w = 0; w < grid * 4; ++w) tmin = h[2 * w + 1] < tmin ? h[2 * w + 1] : tmin;
    printf("priorities of workgroup classes 0/1/2 (launch order): %d %d %d, %u chain(s); %u x 8 instructions per wave\n", p0, p1, p2, chains, iters);
    for (uint32_t c = 0; c < 3; ++c) {
        double sum = 0, start = 0; uint32_t n = 0;
        for (uint32_t b = c * cus; b < (c + 1) * cus; ++b)
            for (uint32_t w = 0; w < 4; ++w) { sum += (double)h[2 * (b * 4 + w)]; start += (double)(h[2 * (b * 4 + w) + 1] - tmin); ++n; }
        printf("  class %u (prio %d): mean start %.1f us, mean time %.1f us  -> %.2f instructions per us per wave\n", c,
               c == 0 ? p0 : c == 1 ? p1 : p2, start / n / 100.0, sum / n / 100.0, iters * 8.0 / (sum / n / 100.0));
    }
    return 0;
}
