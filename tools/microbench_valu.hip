// Integer VALU issue-rate microbenchmark for gfx950 (used to pick the Poseidon/NTT arithmetic form).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o tools/microbench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

#define KERNEL_BEGIN(name) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
    uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7; \
    uint32_t b = seed * 2654435761u + 12345u; \
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7; \
    for (int it = 0; it < ITERS; ++it) {
#define KERNEL_END \
    } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32); \
}

#define OP32(ins) \
    asm volatile(ins " %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a1) : "v"(b)); \
    asm volatile(ins " %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a3) : "v"(b)); \
    asm volatile(ins " %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a5) : "v"(b)); \
    asm volatile(ins " %0, %0, %1" : "+v"(a6) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a7) : "v"(b));
#define OP32_3(ins) \
    asm volatile(ins " %0, %0, %1, %0" : "+v"(a0) : "v"(b)); asm volatile(ins " %0, %0, %1, %0" : "+v"(a1) : "v"(b)); \
    asm volatile(ins " %0, %0, %1, %0" : "+v"(a2) : "v"(b)); asm volatile(ins " %0, %0, %1, %0" : "+v"(a3) : "v"(b)); \
    asm volatile(ins " %0, %0, %1, %0" : "+v"(a4) : "v"(b)); asm volatile(ins " %0, %0, %1, %0" : "+v"(a5) : "v"(b)); \
    asm volatile(ins " %0, %0, %1, %0" : "+v"(a6) : "v"(b)); asm volatile(ins " %0, %0, %1, %0" : "+v"(a7) : "v"(b));
#define OP64_MAD(ins) \
    asm volatile(ins " %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(a0), "v"(b) : "vcc"); asm volatile(ins " %0, vcc, %1, %2, %0" : "+v"(q1) : "v"(a1), "v"(b) : "vcc"); \
    asm volatile(ins " %0, vcc, %1, %2, %0" : "+v"(q2) : "v"(a2), "v"(b) : "vcc"); asm volatile(ins " %0, vcc, %1, %2, %0" : "+v"(q3) : "v"(a3), "v"(b) : "vcc"); \
    asm volatile(ins " %0, vcc, %1, %2, %0" : "+v"(q4) : "v"(a4), "v"(b) : "vcc"); asm volatile(ins " %0, vcc, %1, %2, %0" : "+v"(q5) : "v"(a5), "v"(b) : "vcc"); \
    asm volatile(ins " %0, vcc, %1, %2, %0" : "+v"(q6) : "v"(a6), "v"(b) : "vcc"); asm volatile(ins " %0, vcc, %1, %2, %0" : "+v"(q7) : "v"(a7), "v"(b) : "vcc");
#define OP64_SHIFT(ins) \
    asm volatile(ins " %0, 3, %0" : "+v"(q0)); asm volatile(ins " %0, 3, %0" : "+v"(q1)); \
    asm volatile(ins " %0, 3, %0" : "+v"(q2)); asm volatile(ins " %0, 3, %0" : "+v"(q3)); \
    asm volatile(ins " %0, 3, %0" : "+v"(q4)); asm volatile(ins " %0, 3, %0" : "+v"(q5)); \
    asm volatile(ins " %0, 3, %0" : "+v"(q6)); asm volatile(ins " %0, 3, %0" : "+v"(q7));

#define R4(x) x x x x
KERNEL_BEGIN(k_add_u32) R4(OP32("v_add_u32")) KERNEL_END
KERNEL_BEGIN(k_xor_b32) R4(OP32("v_xor_b32")) KERNEL_END
KERNEL_BEGIN(k_mul_lo_u32) R4(OP32("v_mul_lo_u32")) KERNEL_END
KERNEL_BEGIN(k_mul_hi_u32) R4(OP32("v_mul_hi_u32")) KERNEL_END
KERNEL_BEGIN(k_mul_u32_u24) R4(OP32("v_mul_u32_u24")) KERNEL_END
KERNEL_BEGIN(k_mad_u32_u24) R4(OP32_3("v_mad_u32_u24")) KERNEL_END
KERNEL_BEGIN(k_mad_u64_u32) R4(OP64_MAD("v_mad_u64_u32")) KERNEL_END
KERNEL_BEGIN(k_lshlrev_b64) R4(OP64_SHIFT("v_lshlrev_b64")) KERNEL_END
KERNEL_BEGIN(k_dot4_u32_u8) R4(OP32_3("v_dot4_u32_u8")) KERNEL_END
KERNEL_BEGIN(k_add3_u32) R4(OP32_3("v_add3_u32")) KERNEL_END
KERNEL_BEGIN(k_lshl_add_u32) R4(OP32_3("v_lshl_add_u32")) KERNEL_END
KERNEL_BEGIN(k_mad_u32_u16) R4(OP32_3("v_mad_u32_u16")) KERNEL_END
KERNEL_BEGIN(k_perm_b32) R4(OP32_3("v_perm_b32")) KERNEL_END
KERNEL_BEGIN(k_alignbit) R4(OP32_3("v_alignbit_b32")) KERNEL_END

KERNEL_BEGIN(k_and_b32) R4(OP32("v_and_b32")) KERNEL_END
KERNEL_BEGIN(k_lshlrev_b32) R4(OP32("v_lshlrev_b32")) KERNEL_END
KERNEL_BEGIN(k_sub_u32) R4(OP32("v_sub_u32")) KERNEL_END
KERNEL_BEGIN(k_add_u32_e64) R4(OP32("v_add_u32_e64")) KERNEL_END
KERNEL_BEGIN(k_max_u32) R4(OP32("v_max_u32")) KERNEL_END
#define OP_ADDCO(dummy) \
    asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a0) : "v"(b) : "vcc"); asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a1) : "v"(b) : "vcc"); \
    asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a2) : "v"(b) : "vcc"); asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a3) : "v"(b) : "vcc"); \
    asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a4) : "v"(b) : "vcc"); asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a5) : "v"(b) : "vcc"); \
    asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a6) : "v"(b) : "vcc"); asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a7) : "v"(b) : "vcc");
KERNEL_BEGIN(k_add_co) R4(OP_ADDCO(0)) KERNEL_END
#define OP_CNDMASK32(dummy) \
    asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b) : ); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b) : ); \
    asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b) : ); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b) : ); \
    asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a4) : "v"(b) : ); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a5) : "v"(b) : ); \
    asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a6) : "v"(b) : ); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a7) : "v"(b) : );
KERNEL_BEGIN(k_cndmask_e32) R4(OP_CNDMASK32(0)) KERNEL_END
#define OP_CNDMASK64(dummy) \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a0) : "v"(b) : ); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a1) : "v"(b) : ); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a2) : "v"(b) : ); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a3) : "v"(b) : ); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a4) : "v"(b) : ); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a5) : "v"(b) : ); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a6) : "v"(b) : ); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a7) : "v"(b) : );
KERNEL_BEGIN(k_cndmask_e64) R4(OP_CNDMASK64(0)) KERNEL_END
#define OP_LSHLADD64(dummy) \
    asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q0)); asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q1)); \
    asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q2)); asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q3)); \
    asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q4)); asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q5)); \
    asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q6)); asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q7));
KERNEL_BEGIN(k_lshl_add_u64) R4(OP_LSHLADD64(0)) KERNEL_END
#define OP_FMA64(dummy) \
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q0)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q1)); \
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q2)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q3)); \
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q4)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q5)); \
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q6)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q7));
KERNEL_BEGIN(k_fma_f64) R4(OP_FMA64(0)) KERNEL_END
KERNEL_BEGIN(k_fma_f32) R4(OP32_3("v_fma_f32")) KERNEL_END

// 64-bit add with carry chain: v_add_co_u32 + v_addc_co_u32 (counted as 2 instructions)
__global__ void __launch_bounds__(256) k_add_u64(uint32_t* out, uint32_t seed) {
    uint64_t q[8]; uint64_t b = seed * 0x9E3779B97F4A7C15ull + threadIdx.x;
    for (int i = 0; i < 8; ++i) q[i] = b * (i + 3);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { q[i] += b; asm volatile("" : "+v"(q[i])); }
    }
    uint64_t r = 0; for (int i = 0; i < 8; ++i) r ^= q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

// Goldilocks modular multiply, compiler-scheduled (the form the product kernels use)
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) {
    uint64_t lo = a * b, hi = __umul64hi(a, b);
    uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    uint64_t t0 = lo - hh; if (lo < hh) t0 -= 0xFFFFFFFFull;
    uint64_t t1 = hl * 0xFFFFFFFFull;
    uint64_t r = t0 + t1; if (r < t1) r += 0xFFFFFFFFull;
    if (r >= 0xFFFFFFFF00000001ull) r -= 0xFFFFFFFF00000001ull;
    return r;
}
__global__ void __launch_bounds__(256) k_gl_mul(uint32_t* out, uint32_t seed) {
    uint64_t q[8]; uint64_t b = seed * 0x9E3779B97F4A7C15ull + threadIdx.x;
    for (int i = 0; i < 8; ++i) q[i] = b * (i + 3);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = gl_mul(q[i], b);
    }
    uint64_t r = 0; for (int i = 0; i < 8; ++i) r ^= q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int ops_per_iter; };

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    const int blocks = cus * 8, threads = 256;
    uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * threads * 4));
    Entry es[] = {
        {"v_add_u32", k_add_u32, 8}, {"v_xor_b32", k_xor_b32, 8}, {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_mul_hi_u32", k_mul_hi_u32, 8},
        {"v_mul_u32_u24", k_mul_u32_u24, 8}, {"v_mad_u32_u24", k_mad_u32_u24, 8}, {"v_mad_u64_u32", k_mad_u64_u32, 8},
        {"v_lshlrev_b64", k_lshlrev_b64, 8}, {"v_dot4_u32_u8", k_dot4_u32_u8, 8}, {"v_add3_u32", k_add3_u32, 8},
        {"v_lshl_add_u32", k_lshl_add_u32, 8}, {"v_mad_u32_u16", k_mad_u32_u16, 8}, {"v_perm_b32", k_perm_b32, 8}, {"v_alignbit_b32", k_alignbit, 8},
        {"v_and_b32", k_and_b32, 8}, {"v_lshlrev_b32", k_lshlrev_b32, 8}, {"v_sub_u32", k_sub_u32, 8}, {"v_add_u32_e64", k_add_u32_e64, 8},
        {"v_max_u32", k_max_u32, 8}, {"v_add_co_u32_e32", k_add_co, 8}, {"v_cndmask_b32_e32", k_cndmask_e32, 8}, {"v_cndmask_b32_e64", k_cndmask_e64, 8},
        {"v_lshl_add_u64", k_lshl_add_u64, 8}, {"v_fma_f64", k_fma_f64, 8}, {"v_fma_f32", k_fma_f32, 8},
        {"add_u64(2 instr)", k_add_u64, 2}, {"gl_mul(modmul)", k_gl_mul, 2},
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (auto& e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 7u + rep);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double lane_ops = (double)blocks * threads * ITERS * (e.ops_per_iter == 2 ? 8 : 32);
        double rate = lane_ops / (best * 1e-3);         // lane-ops / s, whole chip
        double per_cu_clk = rate / cus / 2.4e9;         // lane-ops per CU per (2.4 GHz) clock
        printf("%-20s %8.3f ms  %8.2f Tlane-op/s  %7.1f lane-op/clk/CU (@2.4GHz)  => %5.2f cyc per wave64-instr per SIMD\n",
               e.name, best, rate * 1e-12, per_cu_clk, 64.0 * 4 / per_cu_clk);
    }
    return 0;
}
