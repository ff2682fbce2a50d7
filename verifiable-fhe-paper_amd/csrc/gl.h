// Goldilocks field GF(p), p = 2^64 - 2^32 + 1, and GF(p^2) = F[X]/(X^2 - 7): product arithmetic, host + gfx950.
//
// Role in the reference: the types behind `GoldilocksField` / `QuadraticExtension<GoldilocksField>` selected at
// /root/reference/src/main.rs:33-35 (plonky2_field 0.2.0, pinned by /root/reference/Cargo.lock:396-399).
// Device form: the 64x64->128 product is four v_mad_u64_u32 (measured 4.9 cyc/wave64 on gfx950, profiles/
// r01_microbench_valu.txt -- integer multiply is NOT quarter-rate on CDNA4) followed by the 2^64 = 2^32-1,
// 2^96 = -1 reduction.  "canonical" = value < p; the *_nc forms accept and return any u64 residue.
#pragma once
#include <cstdint>
#include <cstddef>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace gl {
using u32 = uint32_t;
using u64 = uint64_t;

constexpr u64 P = 0xFFFFFFFF00000001ull;
constexpr u64 EPS = 0xFFFFFFFFull;
constexpr u64 GENERATOR = 7;                          // MULTIPLICATIVE_GROUP_GENERATOR == coset_shift()
constexpr u64 TWO_ADIC_GENERATOR = 1753635133440165772ull;  // 7^((p-1)/2^32)
constexpr unsigned TWO_ADICITY = 32;

GL_HD u64 canon(u64 x) { return x >= P ? x - P : x; }

// canonical in, canonical out
GL_HD u64 add(u64 a, u64 b) {
    u64 s = a + b;
    if (s < a || s >= P) s -= P;
    return s;
}
GL_HD u64 sub(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_SUB_PLAIN)
    // the borrow of the 64-bit difference selects the correction -(2^32 - 1) (= +p mod 2^64) directly: 5 VALU, where the compiler's form
    // spends a sixth on a 64-bit compare to find the borrow again
    u32 d0, d1, t;
    asm("v_sub_co_u32_e32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32_e32 %1, vcc, %4, %6, vcc\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "=&v"(d0), "=&v"(d1), "=&v"(t)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc");
    return ((u64)d1 << 32) | d0;
#else
    u64 d = a - b;
    if (a < b) d += P;
    return d;
#endif
}
GL_HD u64 neg(u64 a) { return a ? P - a : 0; }

// any u64 residues in, any u64 residue out (no final conditional subtraction)
GL_HD u64 add_nc(u64 a, u64 b_canonical) {
    u64 s = a + b_canonical;
    if (s < b_canonical) s += EPS;  // wrapped: 2^64 = eps; cannot wrap twice because b < p
    return s;
}

// any u64 residues in, any residue out: the sum can wrap, and the corrected sum can wrap once more (never a third time)
GL_HD u64 add_nn(u64 a, u64 b) {
    u64 s = a + b;
    if (s < a) {
        s += EPS;
        if (s < EPS) s += EPS;
    }
    return s;
}

GL_HD void mul_wide(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t0 = (u64)a0 * b0;
    const u64 t1 = (u64)a1 * b0 + (t0 >> 32);
    const u64 t2 = (u64)a0 * b1 + (u32)t1;
    lo = (t2 << 32) | (u32)t0;
    hi = (u64)a1 * b1 + (t1 >> 32) + (t2 >> 32);
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    lo = (u64)p;
    hi = (u64)(p >> 64);
#endif
}

// reduce a 128-bit value to a u64 residue (not necessarily canonical)
GL_HD u64 reduce128_nc(u64 lo, u64 hi) {
    const u64 hi_hi = hi >> 32, hi_lo = hi & EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;            // borrow: -2^64 = -eps
    const u64 t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1)
    u64 r = t0 + t1;
    if (r < t1) r += EPS;
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__)
// Hand-scheduled gfx950 forms (bit-identical residues are not required between forms; all results are congruent mod p
// and every consumer accepts any u64 residue).  They use fixed scratch registers v80..v87 / s[80:87], declared as
// clobbers, because a 64-bit inline-asm operand cannot name its halves and v_mad_u64_u32 needs aligned pairs.
//   product: 4 v_mad_u64_u32 (the a1*b0 term is accumulated onto a0*b1 with its carry-out kept in an SGPR pair) + 3 adds
//   reduce : u = hi_lo * (2^32-1) + lo as ONE v_mad_u64_u32 with carry-out c; r = u - hi_hi with borrow b;
//            r += (c - b) * (2^32 - 1)  -- neither correction can wrap a second time
// 16 VALU + 2 SALU instead of the 26 VALU hipcc emits for the C form below.
// Scratch registers of the single-product form.  A translation unit whose kernels need few registers of their own (the
// NTT kernels) defines GL_ASM_SCRATCH_LOW before including this header: the scratch block then sits at v24..v31 instead
// of v80..v87, so those kernels are not pushed from ~50 to 96 VGPRs (5 -> 8 waves per SIMD).
#if defined(GL_ASM_SCRATCH_LOW)
#define GL_R0 "v24"
#define GL_R1 "v25"
#define GL_R2 "v26"
#define GL_R3 "v27"
#define GL_R4 "v28"
#define GL_R5 "v29"
#define GL_R6 "v30"
#define GL_R7 "v31"
#define GL_P01 "v[24:25]"
#define GL_P23 "v[26:27]"
#define GL_P45 "v[28:29]"
#define GL_Q0 "v32"
#define GL_Q1 "v33"
#define GL_Q2 "v34"
#define GL_Q3 "v35"
#define GL_Q4 "v36"
#define GL_Q5 "v37"
#define GL_Q6 "v38"
#define GL_Q01 "v[32:33]"
#define GL_Q23 "v[34:35]"
#define GL_Q45 "v[36:37]"
#else
#define GL_R0 "v80"
#define GL_R1 "v81"
#define GL_R2 "v82"
#define GL_R3 "v83"
#define GL_R4 "v84"
#define GL_R5 "v85"
#define GL_R6 "v86"
#define GL_R7 "v87"
#define GL_P01 "v[80:81]"
#define GL_P23 "v[82:83]"
#define GL_P45 "v[84:85]"
#define GL_Q0 "v88"
#define GL_Q1 "v89"
#define GL_Q2 "v90"
#define GL_Q3 "v91"
#define GL_Q4 "v92"
#define GL_Q5 "v93"
#define GL_Q6 "v94"
#define GL_Q01 "v[88:89]"
#define GL_Q23 "v[90:91]"
#define GL_Q45 "v[92:93]"
#endif
// The reduction of the 128-bit value (P01 = low 64 bits, R4 = bits 64..95, R5 = bits 96..127) to outputs %0 / %1, shared by the product forms:
//   u = R4 (2^32 - 1) + P01 as ONE v_mad_u64_u32 with carry-out c;  r = u - R5 with borrow b;  r += (c - b)(2^32 - 1).
// b needs u < R5 < 2^32, i.e. one product in 2^32: the borrow's correction (r -= 2^32 - 1, applied FIRST so that the carry's correction
// cannot wrap) sits behind a wave-level branch that is practically never taken, and the common path pays only the carry's correction:
// 6 VALU where computing both corrections as one signed addend took 8 VALU + 2 SALU (round 5).
#define GL_REDUCE_TAIL(UNIQ)                                                          \
    "v_mad_u64_u32 " GL_P01 ", s[80:81], " GL_R4 ", -1, " GL_P01 "\n\t"                \
    "v_sub_co_u32_e32 " GL_R0 ", vcc, " GL_R0 ", " GL_R5 "\n\t"                         \
    "v_subbrev_co_u32_e32 " GL_R1 ", vcc, 0, " GL_R1 ", vcc\n\t"                        \
    "v_cndmask_b32_e64 " GL_R7 ", 0, -1, s[80:81]\n\t"                                  \
    "s_cbranch_vccz .Lgl_red_" UNIQ "\n\t"                                               \
    "v_cndmask_b32_e64 " GL_R6 ", 0, -1, vcc\n\t"                                       \
    "v_sub_co_u32_e32 " GL_R0 ", vcc, " GL_R0 ", " GL_R6 "\n\t"                         \
    "v_subbrev_co_u32_e32 " GL_R1 ", vcc, 0, " GL_R1 ", vcc\n"                           \
    ".Lgl_red_" UNIQ ":\n\t"                                                             \
    "v_add_co_u32_e32 %0, vcc, " GL_R0 ", " GL_R7 "\n\t"                                \
    "v_addc_co_u32_e64 %1, vcc, " GL_R1 ", 0, vcc"
__device__ __forceinline__ u64 mul_nc(u64 a, u64 b) {
    u32 r0, r1;
    asm("v_mad_u64_u32 " GL_P01 ", vcc, %2, %4, 0\n\t"
        "v_mad_u64_u32 " GL_P23 ", vcc, %2, %5, 0\n\t"
        "v_mad_u64_u32 " GL_P23 ", s[80:81], %3, %4, " GL_P23 "\n\t"
        "v_mad_u64_u32 " GL_P45 ", vcc, %3, %5, 0\n\t"
        "v_cndmask_b32_e64 " GL_R6 ", 0, 1, s[80:81]\n\t"
        "v_add_co_u32_e32 " GL_R1 ", vcc, " GL_R1 ", " GL_R2 "\n\t"
        "v_addc_co_u32_e32 " GL_R4 ", vcc, " GL_R4 ", " GL_R3 ", vcc\n\t"
        "v_addc_co_u32_e32 " GL_R5 ", vcc, " GL_R5 ", " GL_R6 ", vcc\n\t"
        GL_REDUCE_TAIL("%=")
        : "=&v"(r0), "=&v"(r1)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : GL_R0, GL_R1, GL_R2, GL_R3, GL_R4, GL_R5, GL_R6, GL_R7, "vcc", "scc", "s80", "s81");   // only what the form writes: every named scalar is one the allocator loses
    return ((u64)r1 << 32) | r0;
}
// Two independent products with their instruction streams interleaved (second register set v88..v95 / s[86:91]): used
// where a single wave has to hide its own dependent-instruction latency (x^3 and x^4 of the S-box in the 16-lane
// Poseidon).  r = a * b, q = c * d.
__device__ __forceinline__ void mul2_nc(u64 a, u64 b, u64 c, u64 d, u64& r, u64& q) {
    u32 r0, r1, q0, q1;
    asm("v_mad_u64_u32 v[80:81], vcc, %4, %6, 0\n\t"
        "v_mad_u64_u32 v[88:89], vcc, %8, %10, 0\n\t"
        "v_mad_u64_u32 v[82:83], vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 v[90:91], vcc, %8, %11, 0\n\t"
        "v_mad_u64_u32 v[82:83], s[80:81], %5, %6, v[82:83]\n\t"
        "v_mad_u64_u32 v[90:91], s[86:87], %9, %10, v[90:91]\n\t"
        "v_mad_u64_u32 v[84:85], vcc, %5, %7, 0\n\t"
        "v_mad_u64_u32 v[92:93], vcc, %9, %11, 0\n\t"
        "v_cndmask_b32_e64 v86, 0, 1, s[80:81]\n\t"
        "v_cndmask_b32_e64 v94, 0, 1, s[86:87]\n\t"
        "v_add_co_u32_e32 v81, vcc, v81, v82\n\t"
        "v_addc_co_u32_e32 v84, vcc, v84, v83, vcc\n\t"
        "v_addc_co_u32_e32 v85, vcc, v85, v86, vcc\n\t"
        "v_add_co_u32_e32 v89, vcc, v89, v90\n\t"
        "v_addc_co_u32_e32 v92, vcc, v92, v91, vcc\n\t"
        "v_addc_co_u32_e32 v93, vcc, v93, v94, vcc\n\t"
        // both reductions (GL_REDUCE_TAIL): the two borrows share ONE practically-never-taken branch
        "v_mad_u64_u32 v[80:81], s[80:81], v84, -1, v[80:81]\n\t"
        "v_mad_u64_u32 v[88:89], s[86:87], v92, -1, v[88:89]\n\t"
        "v_sub_co_u32_e32 v80, vcc, v80, v85\n\t"
        "v_subbrev_co_u32_e64 v81, s[82:83], 0, v81, vcc\n\t"
        "v_sub_co_u32_e32 v88, vcc, v88, v93\n\t"
        "v_subbrev_co_u32_e64 v89, s[84:85], 0, v89, vcc\n\t"
        "v_cndmask_b32_e64 v87, 0, -1, s[80:81]\n\t"
        "v_cndmask_b32_e64 v95, 0, -1, s[86:87]\n\t"
        "s_or_b64 s[80:81], s[82:83], s[84:85]\n\t"    // (for SCC only; s[80:81] is dead here)
        "s_cbranch_scc0 .Lgl_red2_%=\n\t"
        "v_cndmask_b32_e64 v86, 0, -1, s[82:83]\n\t"
        "v_sub_co_u32_e32 v80, vcc, v80, v86\n\t"
        "v_subbrev_co_u32_e32 v81, vcc, 0, v81, vcc\n\t"
        "v_cndmask_b32_e64 v94, 0, -1, s[84:85]\n\t"
        "v_sub_co_u32_e32 v88, vcc, v88, v94\n\t"
        "v_subbrev_co_u32_e32 v89, vcc, 0, v89, vcc\n"
        ".Lgl_red2_%=:\n\t"
        "v_add_co_u32_e32 %0, vcc, v80, v87\n\t"
        "v_addc_co_u32_e64 %1, vcc, v81, 0, vcc\n\t"
        "v_add_co_u32_e32 %2, vcc, v88, v95\n\t"
        "v_addc_co_u32_e64 %3, vcc, v89, 0, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(q0), "=&v"(q1)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)), "v"((u32)c), "v"((u32)(c >> 32)), "v"((u32)d),
          "v"((u32)(d >> 32))
        : "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "vcc", "scc",
          "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87");
    r = ((u64)r1 << 32) | r0;
    q = ((u64)q1 << 32) | q0;
}
// a * b + c * d with ONE reduction: the two 128-bit products are added (the sum can reach 2^129: its top bit k joins hi_hi) and then
// lo + hi_lo 2^64 + (hi_hi + k 2^32) 2^96 is reduced as in mul_nc.  The subtrahend hi_hi + k 2^32 is below 2^33 instead of 2^32, and
// the single correction (c - b)(2^32 - 1) still cannot wrap: c = 1 means u < 2^64 - 2^33 + 1, so u + 2^32 - 1 < 2^64; b = 1 without c
// means the wrapped difference is at least 2^64 - 2^33, so taking 2^32 - 1 away stays positive.  Any u64 residues in, a residue out.
// 29 VALU + 2 SALU against 2 x 16 + 8 for two products and a modular addition.
__device__ __forceinline__ u64 dot2_nc(u64 a, u64 b, u64 c, u64 d) {
    u32 r0, r1;
    asm("v_mad_u64_u32 " GL_P01 ", vcc, %2, %4, 0\n\t"
        "v_mad_u64_u32 " GL_Q01 ", vcc, %6, %8, 0\n\t"
        "v_mad_u64_u32 " GL_P23 ", vcc, %2, %5, 0\n\t"
        "v_mad_u64_u32 " GL_Q23 ", vcc, %6, %9, 0\n\t"
        "v_mad_u64_u32 " GL_P23 ", s[80:81], %3, %4, " GL_P23 "\n\t"
        "v_mad_u64_u32 " GL_Q23 ", s[86:87], %7, %8, " GL_Q23 "\n\t"
        "v_mad_u64_u32 " GL_P45 ", vcc, %3, %5, 0\n\t"
        "v_mad_u64_u32 " GL_Q45 ", vcc, %7, %9, 0\n\t"
        "v_cndmask_b32_e64 " GL_R6 ", 0, 1, s[80:81]\n\t"
        "v_cndmask_b32_e64 " GL_Q6 ", 0, 1, s[86:87]\n\t"
        "v_add_co_u32_e32 " GL_R1 ", vcc, " GL_R1 ", " GL_R2 "\n\t"
        "v_addc_co_u32_e32 " GL_R4 ", vcc, " GL_R4 ", " GL_R3 ", vcc\n\t"
        "v_addc_co_u32_e32 " GL_R5 ", vcc, " GL_R5 ", " GL_R6 ", vcc\n\t"
        "v_add_co_u32_e32 " GL_Q1 ", vcc, " GL_Q1 ", " GL_Q2 "\n\t"
        "v_addc_co_u32_e32 " GL_Q4 ", vcc, " GL_Q4 ", " GL_Q3 ", vcc\n\t"
        "v_addc_co_u32_e32 " GL_Q5 ", vcc, " GL_Q5 ", " GL_Q6 ", vcc\n\t"
        "v_add_co_u32_e32 " GL_R0 ", vcc, " GL_R0 ", " GL_Q0 "\n\t"
        "v_addc_co_u32_e32 " GL_R1 ", vcc, " GL_R1 ", " GL_Q1 ", vcc\n\t"
        "v_addc_co_u32_e32 " GL_R4 ", vcc, " GL_R4 ", " GL_Q4 ", vcc\n\t"
        "v_addc_co_u32_e32 " GL_R5 ", vcc, " GL_R5 ", " GL_Q5 ", vcc\n\t"
        "v_addc_co_u32_e64 " GL_R6 ", vcc, 0, 0, vcc\n\t"
        // as GL_REDUCE_TAIL with the 33-bit subtrahend R5 + R6 2^32 (the borrow needs u < 2^33: as rare)
        "v_mad_u64_u32 " GL_P01 ", s[80:81], " GL_R4 ", -1, " GL_P01 "\n\t"
        "v_sub_co_u32_e32 " GL_R0 ", vcc, " GL_R0 ", " GL_R5 "\n\t"
        "v_subb_co_u32_e32 " GL_R1 ", vcc, " GL_R1 ", " GL_R6 ", vcc\n\t"
        "v_cndmask_b32_e64 " GL_R7 ", 0, -1, s[80:81]\n\t"
        "s_cbranch_vccz .Lgl_redd_%=\n\t"
        "v_cndmask_b32_e64 " GL_R6 ", 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 " GL_R0 ", vcc, " GL_R0 ", " GL_R6 "\n\t"
        "v_subbrev_co_u32_e32 " GL_R1 ", vcc, 0, " GL_R1 ", vcc\n"
        ".Lgl_redd_%=:\n\t"
        "v_add_co_u32_e32 %0, vcc, " GL_R0 ", " GL_R7 "\n\t"
        "v_addc_co_u32_e64 %1, vcc, " GL_R1 ", 0, vcc"
        : "=&v"(r0), "=&v"(r1)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)), "v"((u32)c), "v"((u32)(c >> 32)), "v"((u32)d),
          "v"((u32)(d >> 32))
        : GL_R0, GL_R1, GL_R2, GL_R3, GL_R4, GL_R5, GL_R6, GL_R7, GL_Q0, GL_Q1, GL_Q2, GL_Q3, GL_Q4, GL_Q5, GL_Q6, "vcc", "scc",
          "s80", "s81", "s86", "s87");
    return ((u64)r1 << 32) | r0;
}
// a * b + c with one reduction (the product plus a 64-bit addend stays below 2^128).  Any residues in, a residue out; 16 VALU.
__device__ __forceinline__ u64 mad_nc(u64 a, u64 b, u64 c) {
    u32 r0, r1;
    // c is the 64-bit addend of the first multiply-add (T = a0 b0 + c, carry kc); kc joins the high half as a carry-in: two adds where
    // adding c to the assembled product took four (round 5).  (The fully chained form -- every partial product taking the previous one's
    // high word as its addend, 6 instead of 8 VALU for the assembly -- needs the register pairs {T1, 0} and {U1, k}, and gfx950 wants VGPR
    // pairs 64-bit aligned: a result's high word always sits in an odd register.)
    asm("v_mad_u64_u32 " GL_P01 ", s[82:83], %2, %4, %6\n\t"
        "v_mad_u64_u32 " GL_P23 ", vcc, %2, %5, 0\n\t"
        "v_mad_u64_u32 " GL_P23 ", s[80:81], %3, %4, " GL_P23 "\n\t"
        "v_mad_u64_u32 " GL_P45 ", vcc, %3, %5, 0\n\t"
        "v_cndmask_b32_e64 " GL_R6 ", 0, 1, s[80:81]\n\t"
        "v_add_co_u32_e32 " GL_R1 ", vcc, " GL_R1 ", " GL_R2 "\n\t"
        "v_addc_co_u32_e32 " GL_R4 ", vcc, " GL_R4 ", " GL_R3 ", vcc\n\t"
        "v_addc_co_u32_e32 " GL_R5 ", vcc, " GL_R5 ", " GL_R6 ", vcc\n\t"
        "v_addc_co_u32_e64 " GL_R4 ", vcc, " GL_R4 ", 0, s[82:83]\n\t"
        "v_addc_co_u32_e64 " GL_R5 ", vcc, " GL_R5 ", 0, vcc\n\t"
        GL_REDUCE_TAIL("%=")
        : "=&v"(r0), "=&v"(r1)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)), "v"(c)
        : GL_R0, GL_R1, GL_R2, GL_R3, GL_R4, GL_R5, GL_R6, GL_R7, "vcc", "scc", "s80", "s81", "s82", "s83");
    return ((u64)r1 << 32) | r0;
}
// lo + hi_lo 2^64 + hi_hi 2^96 (mod p) -> a u64 residue: the reduction tail of mul_nc on its own (6 VALU), for values that are
// 128 bits wide by construction (a field element times a power of two: the shifts inside the radix-16 NTT butterflies)
__device__ __forceinline__ u64 reduce128_asm(u64 lo, u32 hi_lo, u32 hi_hi) {
    u32 r0, r1;
    asm("v_mad_u64_u32 " GL_P01 ", s[80:81], %3, -1, %2\n\t"
        "v_sub_co_u32_e32 " GL_R0 ", vcc, " GL_R0 ", %4\n\t"
        "v_subbrev_co_u32_e32 " GL_R1 ", vcc, 0, " GL_R1 ", vcc\n\t"
        "v_cndmask_b32_e64 " GL_R7 ", 0, -1, s[80:81]\n\t"
        "s_cbranch_vccz .Lgl_red128_%=\n\t"
        "v_cndmask_b32_e64 " GL_R6 ", 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 " GL_R0 ", vcc, " GL_R0 ", " GL_R6 "\n\t"
        "v_subbrev_co_u32_e32 " GL_R1 ", vcc, 0, " GL_R1 ", vcc\n"
        ".Lgl_red128_%=:\n\t"
        "v_add_co_u32_e32 %0, vcc, " GL_R0 ", " GL_R7 "\n\t"
        "v_addc_co_u32_e64 %1, vcc, " GL_R1 ", 0, vcc"
        : "=&v"(r0), "=&v"(r1)
        : "v"(lo), "v"(hi_lo), "v"(hi_hi)
        : GL_R0, GL_R1, GL_R6, GL_R7, "vcc", "scc", "s80", "s81");
    return ((u64)r1 << 32) | r0;
}
// lo + hi 2^64 (mod p) for hi < 2^32: u = hi (2^32 - 1) + lo as one v_mad_u64_u32, its carry-out selects the single +(2^32 - 1) correction
// (the wrapped sum is below hi 2^32, so the corrected sum cannot wrap again).  4 VALU.
__device__ __forceinline__ u64 reduce96_asm(u64 lo, u32 hi) {
    u32 r0, r1;
    asm("v_mad_u64_u32 " GL_P01 ", vcc, %3, -1, %2\n\t"
        "v_cndmask_b32_e64 " GL_R6 ", 0, -1, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, " GL_R0 ", " GL_R6 "\n\t"
        "v_addc_co_u32_e64 %1, vcc, " GL_R1 ", 0, vcc"
        : "=&v"(r0), "=&v"(r1)
        : "v"(lo), "v"(hi)
        : GL_R0, GL_R1, GL_R6, "vcc");
    return ((u64)r1 << 32) | r0;
}
// a + b and a - b for ANY u64 residues, a u64 residue out, always correct: the carry (borrow) of the 64-bit operation selects the correction
// +-(2^32 - 1); the corrected value can wrap a second time only when both operands sit in the top 2^32 of the u64 range (for random data
// once in 2^32 operations), and that case is a wave-level branch to a second correction that is practically never taken.  5 VALU each where
// the canonical forms cost 6 / 5 and need canonical operands -- which costs every product and shift feeding them a 4-instruction canon.
__device__ __forceinline__ u64 add_a(u64 a, u64 b) {
    u32 r0, r1, t;
    asm("v_add_co_u32_e32 %0, vcc, %3, %5\n\t"
        "v_addc_co_u32_e32 %1, vcc, %4, %6, vcc\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_addc_co_u32_e64 %1, vcc, 0, %1, vcc\n\t"
        "s_cbranch_vccz .Lgl_add_a_%=\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_addc_co_u32_e64 %1, vcc, 0, %1, vcc\n"
        ".Lgl_add_a_%=:"
        : "=&v"(r0), "=&v"(r1), "=&v"(t)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc");
    return ((u64)r1 << 32) | r0;
}
__device__ __forceinline__ u64 sub_a(u64 a, u64 b) {
    u32 r0, r1, t;
    asm("v_sub_co_u32_e32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32_e32 %1, vcc, %4, %6, vcc\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "s_cbranch_vccz .Lgl_sub_a_%=\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n"
        ".Lgl_sub_a_%=:"
        : "=&v"(r0), "=&v"(r1), "=&v"(t)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc");
    return ((u64)r1 << 32) | r0;
}
#else
GL_HD u64 add_a(u64 a, u64 b) { return add_nn(a, b); }
GL_HD u64 sub_a(u64 a, u64 b) {
    u64 d = a - b;
    if (a < b) {
        const u64 e = d - EPS;
        d = d < EPS ? e - EPS : e;
    }
    return d;
}
GL_HD u64 mul_nc(u64 a, u64 b) {
    u64 lo, hi;
    mul_wide(a, b, lo, hi);
    return reduce128_nc(lo, hi);
}
GL_HD u64 reduce128_asm(u64 lo, u32 hi_lo, u32 hi_hi) { return reduce128_nc(lo, ((u64)hi_hi << 32) | hi_lo); }
GL_HD u64 reduce96_asm(u64 lo, u32 hi) { return reduce128_nc(lo, hi); }
// host forms of the device's fused products (same residue classes)
GL_HD u64 dot2_nc(u64 a, u64 b, u64 c, u64 d) {
    const unsigned __int128 s = (unsigned __int128)mul_nc(a, b) + mul_nc(c, d);
    return reduce128_nc((u64)s, (u64)(s >> 64));
}
GL_HD u64 mad_nc(u64 a, u64 b, u64 c) {
    const unsigned __int128 s = (unsigned __int128)mul_nc(a, b) + c;
    return reduce128_nc((u64)s, (u64)(s >> 64));
}
GL_HD void mul2_nc(u64 a, u64 b, u64 c, u64 d, u64& r, u64& q) {
    r = mul_nc(a, b);
    q = mul_nc(c, d);
}
#endif
GL_HD u64 mul(u64 a, u64 b) { return canon(mul_nc(a, b)); }

#if defined(__HIPCC__)
// A sum of 64 x 64-bit products with the reduction mod p deferred to the end (device code): the four 32 x 32 partial products of c * a are
// accumulated with v_mad_u64_u32 into three 64-bit lanes at bit offsets 0 / 32 / 64 (their wrap-arounds counted on the side), 8 instructions
// per product instead of a full modular multiply-add (~20-45).  Users: the gate kernels' reduce_with_powers over the constraints (a = a power
// of alpha, wave-uniform: scalar operands, mac) and the FRI sums over polynomials and coefficients (fri.hip: vector operands, mac_v).
struct LazyAcc {
    u64 e, m, h;      // sum of c0*a0 | c0*a1 + c1*a0 | c1*a1   (mod 2^64 each)
    u32 ce, cm, ch;   // wrap-arounds of e, m, h
    __device__ __forceinline__ void mac(u32 c0, u32 c1, u32 a0, u32 a1) {   // a0 / a1 wave-uniform
        asm("v_mad_u64_u32 %0, vcc, %6, %8, %0\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
            "v_mad_u64_u32 %1, vcc, %6, %9, %1\n\t"
            "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
            "v_mad_u64_u32 %1, vcc, %7, %8, %1\n\t"
            "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
            "v_mad_u64_u32 %2, vcc, %7, %9, %2\n\t"
            "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc"
            : "+v"(e), "+v"(m), "+v"(h), "+v"(ce), "+v"(cm), "+v"(ch)
            : "v"(c0), "v"(c1), "s"(a0), "s"(a1)
            : "vcc");
    }
    __device__ __forceinline__ void mac_v(u64 c, u64 a) {   // any two u64 residues, per lane
        asm("v_mad_u64_u32 %0, vcc, %6, %8, %0\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
            "v_mad_u64_u32 %1, vcc, %6, %9, %1\n\t"
            "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
            "v_mad_u64_u32 %1, vcc, %7, %8, %1\n\t"
            "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
            "v_mad_u64_u32 %2, vcc, %7, %9, %2\n\t"
            "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc"
            : "+v"(e), "+v"(m), "+v"(h), "+v"(ce), "+v"(cm), "+v"(ch)
            : "v"((u32)c), "v"((u32)(c >> 32)), "v"((u32)a), "v"((u32)(a >> 32))
            : "vcc");
    }
    // e + 2^32 m + 2^64 h + 2^64 ce + 2^96 cm + 2^128 ch  (mod p), a u64 residue;  2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32.
    // The six pieces are first added as ONE 160-bit integer lo + H_lo 2^64 + hh 2^96 + top 2^128 (plain carries), which then takes a single
    // 128-bit reduction and one subtraction -- a third of the instructions of reducing the pieces one by one (round 5).  Good for up to 2^31
    // products (top, the count of wrap-arounds, is shifted by 32).
    __device__ __forceinline__ u64 reduce() const {
        const u64 lo = e + (m << 32);
        const u64 t = (m >> 32) + ce + (lo < e ? 1u : 0u);     // < 2^34
        const u64 H = h + t;
        const u64 hh = (H >> 32) + cm;                          // < 2^33
        const u64 top = (u64)ch + (H < t ? 1u : 0u) + (hh >> 32);   // wrap-around counts
        return sub_a(reduce128_asm(lo, (u32)H, (u32)hh), top << 32);
    }
};
#endif

// x * 2^24, x * 2^48, x * 2^72 (canonical in and out).  These are the 8th roots of unity up to sign: w_8 = 2^120 = -2^24,
// w_8^2 = w_4 = 2^48, w_8^3 = -2^72 (2^96 = -1), so a radix-8 NTT butterfly needs shifts, not multiplications, inside.
GL_HD u64 mul_2e24(u64 x) {
    const u64 lo = x << 24, hi = x >> 40;      // x 2^24 = lo + hi 2^64 = lo + hi (2^32 - 1)
    const u64 t = (hi << 32) - hi;             // < 2^56
    u64 r = lo + t;
    if (r < t) r += EPS;                       // wrapped: the true sum was r + 2^64
    return canon(r);
}
GL_HD u64 mul_2e48(u64 x) {
    const u64 lo = x << 48, h = x >> 16;       // x 2^48 = lo + h 2^64, h = h1 2^32 + h0 -> h0 (2^32 - 1) - h1   (2^96 = -1)
    const u64 h0 = h & EPS, h1 = h >> 32;
    const u64 t = (h0 << 32) - h0;             // < 2^64 - 2^33 + 1
    u64 r = lo + t;
    if (r < t) r += EPS;
    r = canon(r);
    return sub(r, h1);
}
// 7 x = 8 x - x (canonical in and out): the coset generator steps k_j = 7^j of the permutation argument
GL_HD u64 mul7(u64 x) {
    const u64 lo = x << 3, hi = x >> 61;       // 8 x = lo + hi 2^64 = lo + hi (2^32 - 1), hi < 8
    const u64 t = (hi << 32) - hi;
    u64 r = lo + t;
    if (r < t) r += EPS;
    return sub(canon(r), x);
}
GL_HD u64 mul_2e72(u64 x) {
    // x 2^72 = (x 2^8) 2^64, x 2^8 = c 2^64 + b 2^32 + a  ->  a (2^32 - 1) - b - c 2^32   (2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32)
    const u64 a = (x << 8) & EPS, b = (x >> 24) & EPS, c = x >> 56;
    const u64 pos = (a << 32) - a;             // <= (2^32 - 1)^2 < p
    return sub(pos, b + (c << 32));            // b + c 2^32 < 2^40
}
GL_HD u64 sqr(u64 a) { return mul(a, a); }

GL_HD u64 pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = mul(r, b);
        b = mul(b, b);
        e >>= 1;
    }
    return r;
}
// a^(p-2) with p - 2 = 2^64 - 2^32 - 1 = (2^32 - 2) 2^32 + (2^32 - 1): an addition chain on the exponents 2^k - 1
// (k = 2, 3, 6, 12, 24, 30, 31, 32) -- 64 squarings + 9 multiplications instead of the 127 of square-and-multiply.  inv(0) = 0.
GL_HD u64 inv(u64 a) {
    auto sqn = [](u64 x, int n) {
        for (int i = 0; i < n; ++i) x = mul_nc(x, x);
        return x;
    };
    const u64 t2 = mul_nc(mul_nc(a, a), a);     // a^(2^2 - 1)
    const u64 t3 = mul_nc(mul_nc(t2, t2), a);   // a^(2^3 - 1)
    const u64 t6 = mul_nc(sqn(t3, 3), t3);
    const u64 t12 = mul_nc(sqn(t6, 6), t6);
    const u64 t24 = mul_nc(sqn(t12, 12), t12);
    const u64 t30 = mul_nc(sqn(t24, 6), t6);
    const u64 t31 = mul_nc(mul_nc(t30, t30), a);
    const u64 t32 = mul_nc(mul_nc(t31, t31), a);  // a^(2^32 - 1)
    return canon(mul_nc(sqn(t31, 33), t32));       // (a^(2^31 - 1))^(2^33) * a^(2^32 - 1)
}
GL_HD u64 root_of_unity(unsigned k) {  // primitive_root_of_unity(k)
    u64 g = TWO_ADIC_GENERATOR;
    for (unsigned i = k; i < TWO_ADICITY; ++i) g = mul(g, g);
    return g;
}

// ---- GF(p^2), W = 7 ----
struct Ext {
    u64 c0, c1;
};
GL_HD Ext ext(u64 a, u64 b = 0) { return Ext{a, b}; }
GL_HD Ext add(Ext a, Ext b) { return Ext{add(a.c0, b.c0), add(a.c1, b.c1)}; }
GL_HD Ext sub(Ext a, Ext b) { return Ext{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
GL_HD Ext mul(Ext a, Ext b) {
    const u64 a1b1 = mul(a.c1, b.c1);
    // 7*x = 8x - x
    const u64 seven = sub(mul(a1b1, 8), a1b1);
    return Ext{add(mul(a.c0, b.c0), seven), add(mul(a.c0, b.c1), mul(a.c1, b.c0))};
}
GL_HD Ext mul(Ext a, u64 s) { return Ext{mul(a.c0, s), mul(a.c1, s)}; }
GL_HD Ext pow(Ext b, u64 e) {
    Ext r = ext(1);
    while (e) {
        if (e & 1) r = mul(r, b);
        b = mul(b, b);
        e >>= 1;
    }
    return r;
}
GL_HD Ext inv(Ext a) {
    const u64 a1sq = mul(a.c1, a.c1);
    const u64 norm = sub(mul(a.c0, a.c0), sub(mul(a1sq, 8), a1sq));
    const u64 ni = inv(norm);
    return Ext{mul(a.c0, ni), mul(neg(a.c1), ni)};
}
GL_HD bool eq(Ext a, Ext b) { return a.c0 == b.c0 && a.c1 == b.c1; }

GL_HD u32 bitrev32(u32 x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    u32 r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}
}  // namespace gl
