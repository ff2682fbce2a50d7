// host-side cost of small stream-ordered H2D copies: pageable vs pinned source (gfx950 box)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long* p, int iters) { unsigned long long x = *p; for (int i = 0; i < iters; ++i) x = x * 6364136223846793005ull + 1; *p = x; }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    void* d; hipMalloc(&d, 1 << 20);
    std::vector<char> pageable(1 << 16, 1);
    void* pinned; hipHostMalloc(&pinned, 1 << 16);
    for (size_t bytes : {64, 2048, 8192, 32768}) {
        for (int mode = 0; mode < 2; ++mode) {
            const void* src = mode ? pinned : (const void*)pageable.data();
            // GPU busy with a ~200 us kernel: how long does the enqueue hold the host?
            double host_us = 0, total_us = 0;
            const int reps = 50;
            for (int r = 0; r < reps; ++r) {
                hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, s, (unsigned long long*)d, 60000);
                auto t0 = std::chrono::steady_clock::now();
                hipMemcpyAsync((char*)d + 4096, src, bytes, hipMemcpyHostToDevice, s);
                auto t1 = std::chrono::steady_clock::now();
                hipStreamSynchronize(s);
                auto t2 = std::chrono::steady_clock::now();
                host_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
                total_us += std::chrono::duration<double, std::micro>(t2 - t0).count();
            }
            printf("%6zu B %-8s enqueue holds host %.1f us (kernel+copy drained after %.1f us)\n", bytes, mode ? "pinned" : "pageable", host_us / reps, total_us / reps);
        }
    }
    return 0;
}
