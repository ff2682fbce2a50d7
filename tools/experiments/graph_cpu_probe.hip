// Does launching a fixed kernel sequence as ONE hipGraph take the HSA runtime thread's CPU time down?  (at 2 CPUs per rank that thread costs
// 8.8 ms of CPU per proof -- more than the library's own threads together.)  100 kernels of ~50 us + one synchronise per "step", as direct
// launches and as a graph replay; prints the CPU time of every thread of the process per step, both ways.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_cpu_probe tools/experiments/graph_cpu_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dirent.h>
#include <map>
#include <string>
#include <unistd.h>
__global__ void flag_kernel(volatile unsigned long* flag, unsigned long seq) {
    __threadfence_system();
    *flag = seq;
}
__global__ void spin_kernel(unsigned long* p, unsigned iters) {
    unsigned long x = p[threadIdx.x + blockIdx.x * blockDim.x];
    for (unsigned i = 0; i < iters; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull;
    p[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
static std::map<int, long> thread_ticks() {
    std::map<int, long> out;
    DIR* d = opendir("/proc/self/task");
    while (dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        char path[96], comm[64];
        std::snprintf(path, sizeof path, "/proc/self/task/%s/stat", e->d_name);
        FILE* f = std::fopen(path, "r");
        if (!f) continue;
        int tid; char st; long v[13]; unsigned long ut, stt;
        if (std::fscanf(f, "%d %63s %c", &tid, comm, &st) == 3) {
            for (int i = 0; i < 10; ++i) std::fscanf(f, "%ld", &v[i]);
            std::fscanf(f, "%lu %lu", &ut, &stt);
            out[tid] = (long)(ut + stt);
        }
        std::fclose(f);
    }
    closedir(d);
    return out;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 300, kernels = 100;
    unsigned long* d;
    CK(hipMalloc(&d, 256 * 1024 * 8));
    CK(hipMemset(d, 1, 256 * 1024 * 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto wait = [&] {   // the library's sleeping wait
        while (hipStreamQuery(s) == hipErrorNotReady) usleep(30);
    };
    auto run_with = [&](const char* name, auto&& step, auto&& wait_fn) {
        for (int i = 0; i < 5; ++i) { step(); wait_fn(); }
        auto before = thread_ticks();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) { step(); wait_fn(); }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        auto after = thread_ticks();
        std::printf("%-8s %.3f ms per step of %d kernels; CPU ms per step by thread:", name, ms / steps, kernels);
        for (auto& kv : after) {
            const long dt = kv.second - (before.count(kv.first) ? before[kv.first] : 0);
            if (dt > 0) std::printf("  %s%d: %.2f", kv.first == getpid() ? "main " : "", kv.first, 10.0 * dt / steps);
        }
        std::printf("\n");
    };
    auto run = [&](const char* name, auto&& step) { run_with(name, step, wait); };
    auto launches = [&] { for (int k = 0; k < kernels; ++k) hipLaunchKernelGGL(spin_kernel, dim3(1024), dim3(256), 0, s, d, 2000u); };
    run("direct", launches);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    launches();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    run("graph", [&] { CK(hipGraphLaunch(ge, s)); });
    run("direct", launches);
    run_with("sync", launches, [&] { CK(hipStreamSynchronize(s)); });
    // completion through a word in host memory the device writes itself: no runtime call while waiting
    volatile unsigned long* flag;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
    *flag = 0;
    unsigned long seq = 0;
    run_with("flag", [&] { launches(); hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, s, flag, ++seq); },
             [&] { while (*flag != seq) usleep(30); });
    run_with("sleep9", launches, [&] { usleep(9000); while (hipStreamQuery(s) == hipErrorNotReady) usleep(30); });
    run("direct", launches);
    return 0;
}
