// Standalone check of the hand-scheduled gfx950 primitives (gl::mul_nc, poseidon::fold96, poseidon::permute)
// against their host C forms.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/test_asm.hip -o tools/test_asm
#include "../verifiable-fhe-paper_amd/csrc/poseidon.h"
#include "../verifiable-fhe-paper_amd/csrc/gates.h"
#include <cstdio>
#include <vector>
#include <random>
using gl::u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_mul(const u64* a, const u64* b, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gl::mul_nc(a[i], b[i]));
}
__global__ void k_mul2(const u64* a, const u64* b, u64* out, int n) {   // the interleaved pair of products (sbox2): both streams, one shared rare branch
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 r, q;
    gl::mul2_nc(a[i], b[i], b[(i + 1) % n], a[(i + 2) % n], r, q);
    out[2 * i] = gl::canon(r);
    out[2 * i + 1] = gl::canon(q);
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
// the non-canonical-residue forms the gate kernels use: add_nn, times7, the lazy algebra product, the fused e t + v p, select_lerp
__global__ void k_addnn(const u64* a, const u64* b, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gl::add_nn(a[i], b[i]));
}
// the any-residue forms of the radix-16 NTT butterflies (round 5): add_a / sub_a with their rare second correction, the shift reductions
__global__ void k_adda(const u64* a, const u64* b, u64* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gl::add_a(a[i], b[i]));
}
__global__ void k_suba(const u64* a, const u64* b, u64* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gl::sub_a(a[i], b[i]));
}
__global__ void k_subc(const u64* a, const u64* b, u64* out, int n) {   // gl::sub: canonical operands only
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::sub(gl::canon(a[i]), gl::canon(b[i]));
}
__global__ void k_red(const u64* a, const u64* b, u64* out, int n) {    // out[2 i] = a + (b mod 2^32) 2^64, out[2 i + 1] = a + b 2^64
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[2 * i] = gl::canon(gl::reduce96_asm(a[i], (uint32_t)b[i]));
        out[2 * i + 1] = gl::canon(gl::reduce128_asm(a[i], (uint32_t)b[i], (uint32_t)(b[i] >> 32)));
    }
}
__global__ void k_times7(const u64* a, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gates::times7(a[i]));
}
__global__ void k_alg(const u64* a, const u64* b, u64* out, int n) {   // out[4 i ..]: mul_lazy (2 words), fma2 (2 words)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gates::Alg<u64> x{a[i], b[i]}, y{b[(i + 3) % n], a[(i + 11) % n]}, v{a[(i + 17) % n], b[(i + 19) % n]}, p{gl::canon(b[(i + 23) % n]), gl::canon(a[(i + 29) % n])};
    const gates::Alg<u64> m = gates::mul_lazy(x, y, gates::times7(y.b));
    out[4 * i] = gl::canon(m.a);
    out[4 * i + 1] = gl::canon(m.b);
    const gates::Alg<u64> f = gates::fma2(x, y, v, p);
    out[4 * i + 2] = gl::canon(f.a);   // residues since round 5 (the evaluators' consumers take any residue)
    out[4 * i + 3] = gl::canon(f.b);
}
__global__ void k_lerp(const u64* a, const u64* b, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(gates::select_lerp(a[i], b[i], a[(i + 5) % n]));   // any residues in, a residue out
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
    // more of the top 2^32 of the u64 range (where a corrected sum or difference wraps a second time) and its neighbourhood
    const u64 edge2[] = {gl::P + 1, ~0ull - 1, 0xFFFFFFFF00000002ull, 0xFFFFFFFEFFFFFFFFull, 0xFFFFFFFF7FFFFFFFull, 0xFFFFFFFFFFFF0000ull, 2, 0xFFFFFFFEull,
                         0x1FFFFFFFFull, 0x7FFFFFFFFFFFFFFFull, 0x8000000000000001ull, 0xFFFFFFFE00000001ull};
    for (int i = 0; i < n; ++i) { a[i] = rng(); b[i] = rng(); }
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) { a[i * 9 + j] = edge[i]; b[i * 9 + j] = edge[j]; }
    {
        u64 all[21];
        for (int i = 0; i < 9; ++i) all[i] = edge[i];
        for (int i = 0; i < 12; ++i) all[9 + i] = edge2[i];
        for (int i = 0; i < 21; ++i) for (int j = 0; j < 21; ++j) { a[128 + i * 21 + j] = all[i]; b[128 + i * 21 + j] = all[j]; }
    }
    u64 *da, *db, *dout;
    CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dout, n * 8));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    int bad = 0, total_bad = 0;
    if (argc < 2 || argv[1][0] == 'm') {
        hipLaunchKernelGGL(k_mul, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            unsigned __int128 p = (unsigned __int128)a[i] * b[i];
            u64 want = (u64)(p % gl::P);
            if (out[i] != want) { if (bad < 5) printf("mul mismatch %d: %016lx * %016lx -> %016lx want %016lx\n", i, a[i], b[i], out[i], want); ++bad; }
        }
        printf("mul_nc: %d mismatches of %d\n", bad, n); total_bad += bad;
        // mul2_nc: the edge pairs reach the borrow's correction in either stream alone and in both (e.g. 2^63 * 2^63 = 2^126: low 96 bits zero)
        u64* dout2;
        CK(hipMalloc(&dout2, 2 * n * 8));
        std::vector<u64> out2(2 * n);
        hipLaunchKernelGGL(k_mul2, dim3(n / 256), dim3(256), 0, 0, da, db, dout2, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out2.data(), dout2, 2 * n * 8, hipMemcpyDeviceToHost));
        bad = 0;
        for (int i = 0; i < n; ++i) {
            const u64 w0 = (u64)(((unsigned __int128)a[i] * b[i]) % gl::P), w1 = (u64)(((unsigned __int128)b[(i + 1) % n] * a[(i + 2) % n]) % gl::P);
            if (out2[2 * i] != w0 || out2[2 * i + 1] != w1) { if (bad < 5) printf("mul2 mismatch %d\n", i); ++bad; }
        }
        printf("mul2_nc: %d mismatches of %d\n", bad, n); total_bad += bad;
        CK(hipFree(dout2));
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
        printf("dot2_nc: %d mismatches of %d\n", bad, n); total_bad += bad;
        bad = 0;
        hipLaunchKernelGGL(k_mad, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const u64 c = a[(i + 5) % n];
            const u64 want = (u64)((((unsigned __int128)a[i] * b[i]) % gl::P + c % gl::P) % gl::P);
            if (out[i] != want) { if (bad < 5) printf("mad mismatch %d: %016lx %016lx %016lx -> %016lx want %016lx\n", i, a[i], b[i], c, out[i], want); ++bad; }
        }
        printf("mad_nc: %d mismatches of %d\n", bad, n); total_bad += bad;
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
        printf("fold96: %d mismatches of %d\n", bad, n); total_bad += bad;
    }
    if (argc < 2 || argv[1][0] == 'g') {
        auto mulm = [](u64 x, u64 y) { return (u64)(((unsigned __int128)(x % gl::P) * (y % gl::P)) % gl::P); };
        auto addm = [](u64 x, u64 y) { return (u64)(((unsigned __int128)(x % gl::P) + (y % gl::P)) % gl::P); };
        CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
        bad = 0;
        hipLaunchKernelGGL(k_addnn, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) if (out[i] != addm(a[i], b[i])) { if (bad < 5) printf("add_nn mismatch %d: %016lx + %016lx -> %016lx\n", i, a[i], b[i], out[i]); ++bad; }
        printf("add_nn: %d mismatches of %d\n", bad, n); total_bad += bad;
        {   // any-residue add / sub (every pair of edge values wraps in some lanes of a wave and not in others), canonical sub, shift reductions
            auto subm = [](u64 x, u64 y) { return (u64)(((unsigned __int128)(x % gl::P) + gl::P - (y % gl::P)) % gl::P); };
            struct { const char* name; void (*k)(const u64*, const u64*, u64*, int); u64 (*want)(u64, u64); } cases[] = {
                {"add_a", k_adda, +[](u64 x, u64 y) { return (u64)(((unsigned __int128)(x % gl::P) + (y % gl::P)) % gl::P); }},
                {"sub_a", k_suba, +[](u64 x, u64 y) { return (u64)(((unsigned __int128)(x % gl::P) + gl::P - (y % gl::P)) % gl::P); }},
                {"sub (asm)", k_subc, +[](u64 x, u64 y) { return (u64)(((unsigned __int128)(x % gl::P) + gl::P - (y % gl::P)) % gl::P); }}};
            (void)subm;
            for (auto& c : cases) {
                bad = 0;
                hipLaunchKernelGGL(c.k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
                for (int i = 0; i < n; ++i) if (out[i] != c.want(a[i], b[i])) { if (bad < 5) printf("%s mismatch %d: %016lx , %016lx -> %016lx\n", c.name, i, a[i], b[i], out[i]); ++bad; }
                printf("%s: %d mismatches of %d\n", c.name, bad, n); total_bad += bad;
            }
            u64* dout2;
            CK(hipMalloc(&dout2, 2 * n * 8));
            std::vector<u64> out2(2 * n);
            hipLaunchKernelGGL(k_red, dim3(n / 256), dim3(256), 0, 0, da, db, dout2, n);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(out2.data(), dout2, 2 * n * 8, hipMemcpyDeviceToHost));
            bad = 0;
            for (int i = 0; i < n; ++i) {
                const unsigned __int128 two64 = (unsigned __int128)1 << 64;
                const u64 w96 = (u64)((a[i] % gl::P + (unsigned __int128)(uint32_t)b[i] * (two64 % gl::P)) % gl::P);
                const u64 w128 = (u64)((a[i] % gl::P + ((unsigned __int128)(b[i] % gl::P) * (two64 % gl::P)) % gl::P) % gl::P);
                if (out2[2 * i] != w96 || out2[2 * i + 1] != w128) { if (bad < 5) printf("reduce mismatch %d: %016lx %016lx -> %016lx %016lx\n", i, a[i], b[i], out2[2 * i], out2[2 * i + 1]); ++bad; }
            }
            printf("reduce96 / reduce128: %d mismatches of %d\n", bad, n); total_bad += bad;
            CK(hipFree(dout2));
        }
        bad = 0;
        hipLaunchKernelGGL(k_times7, dim3(n / 256), dim3(256), 0, 0, da, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) if (out[i] != mulm(a[i], 7)) { if (bad < 5) printf("times7 mismatch %d: %016lx -> %016lx\n", i, a[i], out[i]); ++bad; }
        printf("times7: %d mismatches of %d\n", bad, n); total_bad += bad;
        bad = 0;
        u64* dout4; CK(hipMalloc(&dout4, 4 * (size_t)n * 8));
        std::vector<u64> out4(4 * (size_t)n);
        hipLaunchKernelGGL(k_alg, dim3(n / 256), dim3(256), 0, 0, da, db, dout4, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out4.data(), dout4, 4 * (size_t)n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const u64 xa = a[i], xb = b[i], ya = b[(i + 3) % n], yb = a[(i + 11) % n], va = a[(i + 17) % n], vb = b[(i + 19) % n],
                      pa = b[(i + 23) % n] % gl::P, pb = a[(i + 29) % n] % gl::P;
            const u64 ma = addm(mulm(xa, ya), mulm(7, mulm(xb, yb))), mb = addm(mulm(xa, yb), mulm(xb, ya));
            const u64 qa = addm(mulm(va, pa), mulm(7, mulm(vb, pb))), qb = addm(mulm(va, pb), mulm(vb, pa));
            if (out4[4 * i] != ma || out4[4 * i + 1] != mb || out4[4 * i + 2] != addm(ma, qa) || out4[4 * i + 3] != addm(mb, qb)) {
                if (bad < 5) printf("algebra product mismatch %d\n", i);
                ++bad;
            }
        }
        printf("mul_lazy / fma2: %d mismatches of %d\n", bad, n); total_bad += bad;
        bad = 0;
        hipLaunchKernelGGL(k_lerp, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const u64 x = a[i] % gl::P, y = b[i] % gl::P, sel = a[(i + 5) % n];
            const u64 want = addm(x, mulm(sel, (u64)(((unsigned __int128)y + gl::P - x) % gl::P)));
            if (out[i] != want) { if (bad < 5) printf("select_lerp mismatch %d\n", i); ++bad; }
        }
        printf("select_lerp: %d mismatches of %d\n", bad, n); total_bad += bad;
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
        printf("permute: %d mismatching words of %d\n", bad, 12 * np); total_bad += bad;
    }
    printf("%s\n", total_bad ? "ASM_EDGE_FAILED" : "ASM_EDGE_OK");
    return total_bad ? 1 : 0;
}
