// Microbenchmark: lane-per-permutation vs 16-lane Poseidon kernels across node counts (latency vs throughput regime).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I verifiable-fhe-paper_amd/csrc tools/bench_poseidon.hip -o tools/bench_poseidon
#include "../verifiable-fhe-paper_amd/csrc/hash.hip"
#include <cstdio>
#include <vector>

int main() {
    using namespace vpbs;
    const size_t maxn = (size_t)1 << 20;
    u64 *children, *parents;
    hipMalloc(&children, maxn * 8 * 8);
    hipMalloc(&parents, maxn * 4 * 8);
    hipMemset(children, 1, maxn * 8 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t s; hipStreamCreate(&s);
    printf("%10s %12s %12s %12s\n", "n_parents", "narrow_us", "wide_us", "back2back_gap_us");
    for (size_t n = 16; n <= maxn; n <<= 1) {
        float best[2] = {1e9f, 1e9f};
        for (int variant = 0; variant < 2; ++variant) {
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0, s);
                if (variant == 0)
                    hipLaunchKernelGGL(merkle_level_kernel, dim3((n + 255) / 256), dim3(256), 0, s, children, parents, n);
                else
                    hipLaunchKernelGGL(merkle_level_wide_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, s, children, parents, n);
                hipEventRecord(e1, s);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best[variant]) best[variant] = ms;
            }
        }
        // cost of 8 dependent back-to-back launches of the wide kernel (launch gap estimate)
        hipEventRecord(e0, s);
        for (int k = 0; k < 8; ++k)
            hipLaunchKernelGGL(merkle_level_wide_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, s, children, parents, n);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms8; hipEventElapsedTime(&ms8, e0, e1);
        printf("%10zu %12.1f %12.1f %12.1f\n", n, best[0] * 1e3, best[1] * 1e3, (ms8 / 8 - best[1]) * 1e3);
    }
    return 0;
}
