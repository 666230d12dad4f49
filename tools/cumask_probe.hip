// Probe (not part of the library): do CU-masked streams work on this runtime, and how do mask bits map to CUs?
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
// MI355X, ROCm 7.2: bit i of the mask = CU i/8 of XCD i%8 (bits 0..7 -> one CU on each of the 8 XCDs; the complement ->
// 248 CUs).  Tried for running the batch walk kernel on CUs of its own beside the fill: the masks work, but the fill
// kernel on a masked stream ran 5.5 % slower than the same grid unmasked (49.4 vs 46.9 ms on C5), so the library keeps
// the shared-slot scheme.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void who(uint32_t *out)
{
    uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(64);           // 200 us: keep the slot busy
}
int main()
{
    const int n = 2048;
    uint32_t *d; hipMalloc(&d, n * 8);
    for (int variant = 0; variant < 3; ++variant) {
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) mask[i] = 0;
        if (variant == 0) for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
        if (variant == 1) mask[0] = 0xffu;                 // bits 0..7
        if (variant == 2) { for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu; mask[0] = 0xffffff00u; }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        printf("variant %d create: %s\n", variant, hipGetErrorString(e));
        if (e != hipSuccess) continue;
        hipMemsetAsync(d, 0xff, n * 8, s);
        hipLaunchKernelGGL(who, dim3(n), dim3(64), 0, s, d);
        e = hipStreamSynchronize(s);
        printf("  sync: %s\n", hipGetErrorString(e));
        std::vector<uint32_t> h(2 * n);
        hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        std::set<uint32_t> cus; std::set<uint32_t> xccs;
        for (int i = 0; i < n; ++i) {
            uint32_t hw = h[2 * i], x = h[2 * i + 1] & 0xf;
            uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cus.insert((x << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(x);
        }
        printf("  distinct (xcc,se,sh,cu): %zu  xccs: %zu  sample hw %08x xcc %08x\n", cus.size(), xccs.size(), h[0], h[1]);
        if (cus.size() <= 16) { for (auto c : cus) printf("   %05x", c); printf("\n"); }
        hipStreamDestroy(s);
    }
    return 0;
}
