// host_link_probe.hip -- what the host <-> HBM hand-over of a batch costs on this box, piece by piece.
//
// aln_align_batch takes HOST buffers (the reference's callers own Vec<T>s: statistics/mod.rs:255-286) and returns host
// buffers; this probe measures the building blocks its staging pipeline is designed from:
//   pinned H2D / D2H rate, pageable H2D / D2H rate, hipHostRegister / Unregister cost, host memcpy rate pageable -> pinned
//   with 1..16 threads, hipMalloc / hipFree / hipHostMalloc latency, small-copy and launch latency.
// Build: hipcc --offload-arch=gfx950 -O2 -pthread -o tools/bin/host_link_probe tools/host_link_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHK(x)                                                                                     \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void empty_kernel(uint32_t *p) { if (p && threadIdx.x == 1234567) *p = 0; }

static void par_memcpy(uint8_t *dst, const uint8_t *src, size_t n, int threads)
{
    std::vector<std::thread> th;
    const size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const size_t lo = std::min(n, per * t), hi = std::min(n, lo + per);
        th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto &t : th) t.join();
}

int main(int argc, char **argv)
{
    const size_t MB = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    const size_t n = MB << 20;
    printf("# %zu MiB buffers; host threads available: %u\n", MB, std::thread::hardware_concurrency());
    uint8_t *dev, *pin, *pin2;
    double t0 = now();
    CHK(hipMalloc(&dev, n));
    printf("hipMalloc %zu MiB                 %8.3f ms\n", MB, (now() - t0) * 1e3);
    t0 = now();
    CHK(hipHostMalloc(&pin, n, hipHostMallocDefault));
    printf("hipHostMalloc %zu MiB             %8.3f ms\n", MB, (now() - t0) * 1e3);
    CHK(hipHostMalloc(&pin2, n, hipHostMallocDefault));
    uint8_t *page = (uint8_t *)malloc(n), *page2 = (uint8_t *)malloc(n);
    t0 = now();
    memset(page, 1, n);
    printf("first-touch memset pageable          %8.3f ms  (%.1f GB/s)\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
    memset(page2, 2, n);
    memset(pin, 3, n);
    memset(pin2, 4, n);
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        CHK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, s));
        CHK(hipStreamSynchronize(s));
        double dt = now() - t0;
        printf("pinned   H2D                         %8.3f ms  (%.1f GB/s)\n", dt * 1e3, n / dt / 1e9);
        t0 = now();
        CHK(hipMemcpyAsync(pin2, dev, n, hipMemcpyDeviceToHost, s));
        CHK(hipStreamSynchronize(s));
        dt = now() - t0;
        printf("pinned   D2H                         %8.3f ms  (%.1f GB/s)\n", dt * 1e3, n / dt / 1e9);
        t0 = now();
        CHK(hipMemcpy(dev, page, n, hipMemcpyHostToDevice));
        dt = now() - t0;
        printf("pageable H2D (hipMemcpy)             %8.3f ms  (%.1f GB/s)\n", dt * 1e3, n / dt / 1e9);
        t0 = now();
        CHK(hipMemcpy(page2, dev, n, hipMemcpyDeviceToHost));
        dt = now() - t0;
        printf("pageable D2H (hipMemcpy)             %8.3f ms  (%.1f GB/s)\n", dt * 1e3, n / dt / 1e9);
    }
    // both directions at once on two streams
    {
        hipStream_t s2;
        CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        uint8_t *dev2;
        CHK(hipMalloc(&dev2, n));
        t0 = now();
        CHK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, s));
        CHK(hipMemcpyAsync(pin2, dev2, n, hipMemcpyDeviceToHost, s2));
        CHK(hipStreamSynchronize(s));
        CHK(hipStreamSynchronize(s2));
        double dt = now() - t0;
        printf("pinned H2D + D2H concurrently        %8.3f ms  (%.1f GB/s each way)\n", dt * 1e3, n / dt / 1e9);
        CHK(hipFree(dev2));
        CHK(hipStreamDestroy(s2));
    }
    // register the caller's memory instead of bouncing
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        CHK(hipHostRegister(page, n, hipHostRegisterDefault));
        double dt = now() - t0;
        printf("hipHostRegister %zu MiB            %8.3f ms  (%.1f GB/s)\n", MB, dt * 1e3, n / dt / 1e9);
        t0 = now();
        CHK(hipMemcpyAsync(dev, page, n, hipMemcpyHostToDevice, s));
        CHK(hipStreamSynchronize(s));
        dt = now() - t0;
        printf("registered H2D                       %8.3f ms  (%.1f GB/s)\n", dt * 1e3, n / dt / 1e9);
        t0 = now();
        CHK(hipMemcpyAsync(page, dev, n, hipMemcpyDeviceToHost, s));
        CHK(hipStreamSynchronize(s));
        dt = now() - t0;
        printf("registered D2H                       %8.3f ms  (%.1f GB/s)\n", dt * 1e3, n / dt / 1e9);
        t0 = now();
        CHK(hipHostUnregister(page));
        printf("hipHostUnregister                    %8.3f ms\n", (now() - t0) * 1e3);
    }
    // host memcpy pageable <-> pinned
    for (int threads : {1, 2, 4, 8, 16}) {
        t0 = now();
        par_memcpy(pin, page, n, threads);
        double dt = now() - t0;
        t0 = now();
        par_memcpy(page2, pin2, n, threads);
        double dt2 = now() - t0;
        printf("memcpy %2d thread(s): pageable->pinned %7.3f ms (%.1f GB/s)   pinned->pageable %7.3f ms (%.1f GB/s)\n", threads,
               dt * 1e3, n / dt / 1e9, dt2 * 1e3, n / dt2 / 1e9);
    }
    // many small copies: 100 000 strings of ~1.5 KB scattered (the fetch side's scatter into the caller's layout)
    {
        const size_t cnt = 100000, len = 1500, stride = 4404;
        if (cnt * stride <= n) {
            for (int threads : {1, 4, 8}) {
                t0 = now();
                std::vector<std::thread> th;
                for (int t = 0; t < threads; ++t)
                    th.emplace_back([=] {
                        for (size_t i = t; i < cnt; i += threads) memcpy(page2 + i * stride, pin2 + i * len, len);
                    });
                for (auto &t : th) t.join();
                double dt = now() - t0;
                printf("scatter 100k x 1500 B, %d thread(s)    %8.3f ms  (%.1f GB/s)\n", threads, dt * 1e3, cnt * len / dt / 1e9);
            }
        }
    }
    // latencies
    {
        uint8_t *tmp;
        t0 = now();
        for (int i = 0; i < 20; ++i) { CHK(hipMalloc(&tmp, 64 << 20)); CHK(hipFree(tmp)); }
        printf("hipMalloc+hipFree 64 MiB             %8.3f ms each\n", (now() - t0) * 1e3 / 20);
        t0 = now();
        for (int i = 0; i < 20; ++i) { CHK(hipMalloc(&tmp, 4096)); CHK(hipFree(tmp)); }
        printf("hipMalloc+hipFree 4 KiB              %8.3f ms each\n", (now() - t0) * 1e3 / 20);
        t0 = now();
        for (int i = 0; i < 200; ++i) { CHK(hipMemcpyAsync(dev, pin, 4096, hipMemcpyHostToDevice, s)); CHK(hipStreamSynchronize(s)); }
        printf("pinned 4 KiB H2D + sync              %8.3f us each\n", (now() - t0) * 1e6 / 200);
        t0 = now();
        for (int i = 0; i < 200; ++i) { CHK(hipMemcpyAsync(pin2, dev, 4096, hipMemcpyDeviceToHost, s)); CHK(hipStreamSynchronize(s)); }
        printf("pinned 4 KiB D2H + sync              %8.3f us each\n", (now() - t0) * 1e6 / 200);
        t0 = now();
        for (int i = 0; i < 200; ++i) { CHK(hipMemcpy(dev, page, 4096, hipMemcpyHostToDevice)); }
        printf("pageable 4 KiB H2D (hipMemcpy)       %8.3f us each\n", (now() - t0) * 1e6 / 200);
        t0 = now();
        for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (uint32_t *)nullptr); CHK(hipStreamSynchronize(s)); }
        printf("empty kernel launch + sync           %8.3f us each\n", (now() - t0) * 1e6 / 200);
        t0 = now();
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (uint32_t *)nullptr);
        CHK(hipStreamSynchronize(s));
        printf("200 empty kernels back to back       %8.3f us each\n", (now() - t0) * 1e6 / 200);
        // device-mapped pinned memory: a kernel-visible flag the host polls (no sync call)
        volatile uint32_t *flag;
        CHK(hipHostMalloc((void **)&flag, 4096, hipHostMallocMapped));
        *flag = 1;
        t0 = now();
        for (int i = 0; i < 200; ++i) {
            *flag = 1;
            hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (uint32_t *)nullptr);
            CHK(hipMemsetAsync((void *)flag, 0, 4, s));
            while (*flag) { }
        }
        printf("launch + memset(mapped flag) + poll  %8.3f us each\n", (now() - t0) * 1e6 / 200);
    }
    return 0;
}
