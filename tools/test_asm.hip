// Standalone check of the hand-scheduled gfx950 primitives (gl::mul_nc, poseidon::fold96, poseidon::permute)
// against their host C forms.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/test_asm.hip -o tools/test_asm
#include "../verifiable-fhe-paper_amd/csrc/poseidon.h"
#include <cstdio>
#include <vector>
#include <random>
using gl::u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_mul(const u64* a, const u64* b, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gl::mul_nc(a[i], b[i]));
}
__global__ void k_dot2(const u64* a, const u64* b, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gl::dot2_nc(a[i], b[i], b[(i + 7) % n], a[(i + 13) % n]));
}
__global__ void k_mad(const u64* a, const u64* b, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gl::mad_nc(a[i], b[i], a[(i + 5) % n]));
}
__global__ void k_fold(const u64* a, const u64* b, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(poseidon::fold96(a[i], b[i]));
}
__global__ void k_perm(u64* st, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
    for (int k = 0; k < 12; ++k) s[k] = st[12 * i + k];
    poseidon::permute(s);
    for (int k = 0; k < 12; ++k) st[12 * i + k] = s[k];
}

int main(int argc, char** argv) {
    const int n = 1 << 16;
    std::mt19937_64 rng(1);
    std::vector<u64> a(n), b(n), out(n);
    const u64 edge[] = {0, 1, gl::P - 1, gl::P, ~0ull, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 1ull << 63};
    for (int i = 0; i < n; ++i) { a[i] = rng(); b[i] = rng(); }
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) { a[i * 9 + j] = edge[i]; b[i * 9 + j] = edge[j]; }
    u64 *da, *db, *dout;
    CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dout, n * 8));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    int bad = 0;
    if (argc < 2 || argv[1][0] == 'm') {
        hipLaunchKernelGGL(k_mul, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            unsigned __int128 p = (unsigned __int128)a[i] * b[i];
            u64 want = (u64)(p % gl::P);
            if (out[i] != want) { if (bad < 5) printf("mul mismatch %d: %016lx * %016lx -> %016lx want %016lx\n", i, a[i], b[i], out[i], want); ++bad; }
        }
        printf("mul_nc: %d mismatches of %d\n", bad, n);
    }
    if (argc < 2 || argv[1][0] == 'd') {   // the fused products of the gate kernels: a b + c d and a b + c with one reduction
        bad = 0;
        hipLaunchKernelGGL(k_dot2, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const u64 c = b[(i + 7) % n], d = a[(i + 13) % n];
            const u64 want = (u64)((((unsigned __int128)a[i] * b[i]) % gl::P + ((unsigned __int128)c * d) % gl::P) % gl::P);
            if (out[i] != want) { if (bad < 5) printf("dot2 mismatch %d: %016lx %016lx %016lx %016lx -> %016lx want %016lx\n", i, a[i], b[i], c, d, out[i], want); ++bad; }
        }
        printf("dot2_nc: %d mismatches of %d\n", bad, n);
        bad = 0;
        hipLaunchKernelGGL(k_mad, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const u64 c = a[(i + 5) % n];
            const u64 want = (u64)((((unsigned __int128)a[i] * b[i]) % gl::P + c % gl::P) % gl::P);
            if (out[i] != want) { if (bad < 5) printf("mad mismatch %d: %016lx %016lx %016lx -> %016lx want %016lx\n", i, a[i], b[i], c, out[i], want); ++bad; }
        }
        printf("mad_nc: %d mismatches of %d\n", bad, n);
    }
    if (argc < 2 || argv[1][0] == 'f') {
        bad = 0;
        std::vector<u64> fa(n), fb(n);
        for (int i = 0; i < n; ++i) { fa[i] = rng() >> 6; fb[i] = rng() >> 6; }
        fa[0] = fb[0] = (1ull << 58) - 1; fa[1] = 0; fb[1] = (1ull << 58) - 1; fa[2] = ~0ull >> 6; fb[2] = 0xFFFFFFFFull;
        CK(hipMemcpy(da, fa.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, fb.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fold, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            unsigned __int128 v = (unsigned __int128)fa[i] + ((unsigned __int128)fb[i] << 32);
            u64 want = (u64)(v % gl::P);
            if (out[i] != want) { if (bad < 5) printf("fold mismatch %d: %016lx %016lx -> %016lx want %016lx\n", i, fa[i], fb[i], out[i], want); ++bad; }
        }
        printf("fold96: %d mismatches of %d\n", bad, n);
    }
    if (argc < 2 || argv[1][0] == 'p') {
        bad = 0;
        const int np = 4096;
        std::vector<u64> st(12 * np), ref(12 * np);
        for (auto& x : st) x = rng() % gl::P;
        ref = st;
        for (int i = 0; i < np; ++i) poseidon::permute(&ref[12 * i]);
        u64* dst; CK(hipMalloc(&dst, 12 * np * 8));
        CK(hipMemcpy(dst, st.data(), 12 * np * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_perm, dim3(np / 256), dim3(256), 0, 0, dst, np);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st.data(), dst, 12 * np * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < 12 * np; ++i) if (st[i] != ref[i]) ++bad;
        printf("permute: %d mismatching words of %d\n", bad, 12 * np);
    }
    return 0;
}
