// VALU issue-rate microbenchmark for gfx950, second version (round 2; answers VERDICT r01 "what's weak" #2).
//
// What differs from tools/microbench_valu.hip:
//   * every chain has its OWN source registers (no instruction reads one VGPR for two operands, no operand shared between chains);
//   * the asm statements are not `volatile` (the compiler may schedule them; each result is consumed at the end);
//   * the time is taken INSIDE the kernel, per wave, with s_memtime (shader-clock counter) and s_memrealtime (constant 100 MHz):
//     cycles per instruction need no assumed clock, and the shader clock under the load is reported;
//   * the number of waves per SIMD is controlled exactly (1, 2, 4, 8): one workgroup per CU pinned by its LDS request, 256 * w threads
//     (w <= 4) or two 1024-thread workgroups (w = 8); a workgroup's waves go round-robin over the CU's four SIMDs.
// Reported: cycles of one SIMD per wave64 instruction = (cycles one wave needed) / (instructions of one wave * waves on its SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu2.hip -o tools/microbench_valu2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <array>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int CHAINS = 8;      // independent dependency chains per lane
constexpr int REPS = 4;        // each chain advanced REPS times per loop iteration -> CHAINS * REPS instructions per iteration

struct Stamp {
    uint64_t c0, c1, r0, r1;
    uint32_t hw_id, xcc_id;   // HW_REG_HW_ID (wave slot, SIMD, CU, SH, SE) and HW_REG_XCC_ID: which SIMD of the chip ran the wave
};

extern __shared__ uint32_t lds_pin[];

#define PROLOGUE \
    uint32_t a[CHAINS], b[CHAINS], c[CHAINS], d[CHAINS]; \
    uint64_t q[CHAINS], q2[CHAINS], q3[CHAINS]; \
    for (int i = 0; i < CHAINS; ++i) { \
        a[i] = seed * (2 * i + 3) + threadIdx.x; b[i] = seed * (2 * i + 5) + 7 * threadIdx.x + 1; c[i] = (seed ^ (0x9E3779B9u * (i + 1))) + threadIdx.x; d[i] = c[i] * 3 + b[i]; \
        q[i] = ((uint64_t)a[i] << 32) | b[i]; q2[i] = ((uint64_t)b[i] << 32) | c[i]; q3[i] = ((uint64_t)c[i] << 32) | a[i]; \
    } \
    if (seed == 0xFFFFFFFFu) lds_pin[threadIdx.x] = seed; \
    __syncthreads(); \
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime(); \
    for (int it = 0; it < ITERS; ++it) {
#define EPILOGUE \
    } \
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    uint32_t acc = 0; \
    for (int i = 0; i < CHAINS; ++i) acc ^= a[i] ^ b[i] ^ c[i] ^ d[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ (uint32_t)q2[i] ^ (uint32_t)q3[i]; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc; \
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c0, c1, r0, r1, __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11))};

#define EACH(...) _Pragma("unroll") for (int r = 0; r < REPS; ++r) { _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { __VA_ARGS__; } }

#define KERNEL(name, ...) \
    __global__ void __launch_bounds__(1024) name(uint32_t* out, Stamp* stamps, uint32_t seed) { PROLOGUE EACH(__VA_ARGS__) EPILOGUE }

// a = b * c + a   (three distinct VGPRs per chain)
KERNEL(k_fma_f32, asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i])))
// packed fp32: 2 FMAs per lane per instruction on register pairs
KERNEL(k_pk_fma_f32, asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q[i]) : "v"(q2[i]), "v"(q3[i])))
KERNEL(k_fma_f64, asm("v_fma_f64 %0, %0, %0, %0" : "+v"(q[i])))
KERNEL(k_add_u32, asm("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])))
KERNEL(k_xor_b32, asm("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])))
KERNEL(k_and_or_b32, asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i])))
KERNEL(k_add3_u32, asm("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i])))
KERNEL(k_mul_lo_u32, asm("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])))
KERNEL(k_mul_hi_u32, asm("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])))
KERNEL(k_mad_u32_u24, asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i])))
// carry-writing instructions: chain i sends its carry-out to its own SGPR pair s[20+2i:21+2i], so that no two neighbouring instructions
// write the same SGPR (hipcc separates such a pair with an `s_nop 0`, which costs an issue slot and would be measured along)
#define SP(i, M) M(0, "s[40:41]", "s40", "s41", "s[56:57]", "s56", "s57") M(1, "s[42:43]", "s42", "s43", "s[58:59]", "s58", "s59") M(2, "s[44:45]", "s44", "s45", "s[60:61]", "s60", "s61") M(3, "s[46:47]", "s46", "s47", "s[62:63]", "s62", "s63") M(4, "s[48:49]", "s48", "s49", "s[64:65]", "s64", "s65") M(5, "s[50:51]", "s50", "s51", "s[66:67]", "s66", "s67") M(6, "s[52:53]", "s52", "s53", "s[68:69]", "s68", "s69") M(7, "s[54:55]", "s54", "s55", "s[70:71]", "s70", "s71")
#define KERNEL8(name, M) \
    __global__ void __launch_bounds__(1024) name(uint32_t* out, Stamp* stamps, uint32_t seed) { \
        PROLOGUE _Pragma("unroll") for (int r = 0; r < REPS; ++r) { SP(0, M) } EPILOGUE }
#define M_MAD(i, S, S0, S1, T, T0, T1) asm("v_mad_u64_u32 %0, " S ", %1, %2, %0" : "+v"(q[i]) : "v"(b[i]), "v"(c[i]) : S0, S1);
#define M_MAD_S(i, S, S0, S1, T, T0, T1) asm("v_mad_u64_u32 %0, " S ", %1, %2, %0" : "+v"(q[i]) : "v"(b[i]), "s"(seed) : S0, S1);
#define M_ADDCO(i, S, S0, S1, T, T0, T1) asm("v_add_co_u32_e64 %0, " S ", %1, %0" : "+v"(a[i]) : "v"(b[i]) : S0, S1);
#define M_ADD64(i, S, S0, S1, T, T0, T1) asm("v_add_co_u32_e64 %0, " S ", %2, %0\n\tv_addc_co_u32_e64 %1, " S ", %3, %1, " S : "+v"(a[i]), "+v"(c[i]) : "v"(b[i]), "v"(d[i]) : S0, S1);
#define M_MIX(i, S, S0, S1, T, T0, T1) M_MAD(i, S, S0, S1, T, T0, T1) asm("v_add_co_u32_e64 %0, " T ", %1, %0" : "+v"(a[i]) : "v"(d[i]) : T0, T1);
KERNEL8(k_mad_u64_u32, M_MAD)
// the same with a scalar multiplier (the MDS layer's form: coefficient in an SGPR)
KERNEL8(k_mad_u64_u32_s, M_MAD_S)
KERNEL8(k_add_co_u32, M_ADDCO)
// 64-bit add = v_add_co_u32 + v_addc_co_u32 (2 instructions, counted as 2)
KERNEL8(k_add_u64, M_ADD64)
// the mix of the Poseidon MDS / modular-multiply code: half v_mad_u64_u32, half carry adds
KERNEL8(k_mix_poseidon, M_MIX)
KERNEL(k_lshl_add_u64, asm("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 3) % CHAINS])))
KERNEL(k_cndmask, { asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"((uint64_t)seed * 0x9E3779B97F4A7C15ull)); })
KERNEL(k_lshlrev_b64, asm("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i])))
KERNEL(k_alignbit, asm("v_alignbit_b32 %0, %1, %0, 7" : "+v"(a[i]) : "v"(b[i])))
typedef void (*kern_t)(uint32_t*, Stamp*, uint32_t);
struct Entry {
    const char* name;
    kern_t k;
    int instr_per_body;   // wave instructions per chain step
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %d kHz, LDS per CU %zu\n", prop.gcnArchName, cus, prop.clockRate, prop.maxSharedMemoryPerMultiProcessor);
    const Entry es[] = {
        {"v_fma_f32", k_fma_f32, 1}, {"v_pk_fma_f32 (2 FMA/lane)", k_pk_fma_f32, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_add_u32", k_add_u32, 1},
        {"v_xor_b32", k_xor_b32, 1}, {"v_and_or_b32", k_and_or_b32, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1},
        {"v_mul_hi_u32", k_mul_hi_u32, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mad_u64_u32 (vgpr x vgpr)", k_mad_u64_u32, 1},
        {"v_mad_u64_u32 (vgpr x sgpr)", k_mad_u64_u32_s, 1}, {"v_add_co_u32_e64", k_add_co_u32, 1}, {"add_u64 (add_co+addc)", k_add_u64, 2},
        {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_cndmask_b32_e64", k_cndmask, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1},
        {"v_alignbit_b32", k_alignbit, 1}, {"mix: mad_u64_u32 + add_co", k_mix_poseidon, 2},
    };
    const int ws[] = {1, 2, 4, 8};
    const size_t max_threads = (size_t)cus * 2048;
    uint32_t* d_out;
    Stamp* d_st;
    CHECK(hipMalloc(&d_out, max_threads * 4));
    CHECK(hipMalloc(&d_st, max_threads / 64 * sizeof(Stamp)));
    std::vector<Stamp> st(max_threads / 64);
    hipEvent_t ev0, ev1;
    CHECK(hipEventCreate(&ev0));
    CHECK(hipEventCreate(&ev1));
    printf("cycles of one SIMD per wave64 instruction: in-kernel span of the SIMD's waves / from the HIP-event time of the launch\n");
    printf("%-30s", "");
    for (int w : ws) printf("  w=%d per SIMD", w);
    printf("   sclk(GHz)\n");
    for (const Entry& e : es) {
        printf("%-30s", e.name);
        double ghz = 0;
        for (int w : ws) {
            // one workgroup per CU (w <= 4) or two (w = 8): the LDS request makes a second (third) workgroup not fit
            const int threads = w <= 4 ? 256 * w : 1024;
            const int per_cu = w <= 4 ? 1 : 2;
            const size_t lds = w <= 4 ? 96 * 1024 : 64 * 1024;
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(e.k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int blocks = cus * per_cu;
            double best = 1e30, evt = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(ev0));
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), lds, 0, d_out, d_st, 7u + rep);
                CHECK(hipEventRecord(ev1));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, ev0, ev1));
                const size_t waves = (size_t)blocks * threads / 64;
                CHECK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
                // group the waves by the SIMD that ran them (XCC, SE, SH, CU, SIMD): one SIMD's throughput = the instructions of all its
                // waves / the span from its first start to its last end (s_memtime of one SIMD's waves is one clock domain); waves of a
                // SIMD need not run in lock-step (issue arbitration prefers the older wave), so a per-wave time would mislead
                std::map<uint64_t, std::array<double, 4>> simd;   // key -> {min c0, max c1, waves, sum of per-wave clock}
                for (size_t i = 0; i < waves; ++i) {
                    const uint32_t h = st[i].hw_id;
                    const uint64_t key = ((uint64_t)(st[i].xcc_id & 0xF) << 32) | (h & 0xFF30);   // se_id[15:13] sh_id[12] cu_id[11:8] simd_id[5:4]
                    auto it = simd.find(key);
                    const double clk = (double)(st[i].c1 - st[i].c0) / ((double)(st[i].r1 - st[i].r0) * 10.0);   // s_memrealtime: 100 MHz
                    if (it == simd.end()) simd[key] = {(double)st[i].c0, (double)st[i].c1, 1.0, clk};
                    else {
                        it->second[0] = std::min(it->second[0], (double)st[i].c0);
                        it->second[1] = std::max(it->second[1], (double)st[i].c1);
                        it->second[2] += 1.0;
                        it->second[3] += clk;
                    }
                }
                std::vector<double> per, clks;
                size_t uneven = 0;
                for (auto& kv : simd) {
                    if ((int)kv.second[2] != w) ++uneven;
                    per.push_back((kv.second[1] - kv.second[0]) / ((double)ITERS * CHAINS * REPS * e.instr_per_body * kv.second[2]));
                    clks.push_back(kv.second[3] / kv.second[2]);
                }
                std::nth_element(per.begin(), per.begin() + per.size() / 2, per.end());
                std::nth_element(clks.begin(), clks.begin() + clks.size() / 2, clks.end());
                const double per_instr = per[per.size() / 2];
                if (rep == 1 && (simd.size() != (size_t)cus * 4 || uneven)) printf("[%zu SIMDs seen, %zu without exactly %d waves] ", simd.size(), uneven, w);
                if (rep > 0 && per_instr < best) {
                    best = per_instr;
                    ghz = clks[clks.size() / 2];
                    // cross-check from outside the kernel: HIP-event time of the whole launch (launch overhead and tail included) at the
                    // clock the waves measured
                    evt = ms * 1e-3 * ghz * 1e9 / ((double)ITERS * CHAINS * REPS * e.instr_per_body * w);
                }
            }
            printf("  %5.2f/%5.2f", best, evt);
        }
        printf("   %8.3f\n", ghz);
    }
    return 0;
}
