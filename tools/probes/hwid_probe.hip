// Which wave slots do the three resident waves of a SIMD get?  (Is HW_ID.wave_id usable as a rank among them?)
// hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid_probe tools/probes/hwid_probe.hip && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void probe(uint32_t *out, int spin)
{
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    const uint32_t hw = __builtin_amdgcn_s_getreg((15 << 11) | 4);          // HW_ID[15:0]
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);         // XCC_ID[3:0]
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) out[wave] = hw | (xcc << 16);
    // stay resident so that the whole grid is there at once
    for (volatile int i = 0; i < spin; ++i) { }
}
int main()
{
    const int grid = 768;
    uint32_t *d;
    hipMalloc(&d, grid * 4 * 4);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 50 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(grid * 4);
    hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<int>> by;        // (xcc, se, sh, cu, simd) -> wave ids
    for (int w = 0; w < grid * 4; ++w) by[h[w] >> 4].push_back(h[w] & 15);
    std::map<std::vector<int>, int> hist;
    for (auto &kv : by) { std::sort(kv.second.begin(), kv.second.end()); hist[kv.second]++; }
    printf("%zu SIMDs seen\n", by.size());
    for (auto &kv : hist) { printf("wave ids {"); for (int v : kv.first) printf(" %d", v); printf(" } on %d SIMDs\n", kv.second); }
    // block -> cu mapping of the first blocks
    for (int b = 0; b < 12; ++b) printf("block %d: hw %05x\n", b, h[b * 4]);
    return 0;
}
