// Poseidon permutation over Goldilocks (width 12, rate 8, x^7, 8 full + 22 partial rounds): product code, host +
// gfx950.  Replaces plonky2 0.2.0 hash/poseidon.rs `Poseidon::poseidon` / hash/poseidon_goldilocks.rs as used by
// PoseidonHash (leaf hashing, two_to_one, Challenger); native use in the reference:
// /root/reference/src/vtfhe/ivc_based_vpbs.rs:73.
//
// Arithmetic form (device): the state is kept as arbitrary u64 residues between rounds.  The MDS layer
// (circulant [17,15,41,16,2,28,13,13,39,18,34,20] + diag[8,0..]) is evaluated on the 32-bit halves of the state with
// v_mad_u64_u32 accumulators (all coefficients are inline constants <= 41), the next round's constants are the
// accumulators' initial values (the constant layer costs nothing), and one 96-bit -> 64-bit fold per element ends
// the round.  No MFMA: this is 64-bit modular integer work.
#pragma once
#include "gl.h"

namespace poseidon {
using gl::u32;
using gl::u64;

constexpr int WIDTH = 12, RATE = 8, N_ROUNDS = 30, HALF_FULL = 4, N_PARTIAL = 22;

// 360 round constants + 12 zeros: the constants "after the last round" (the last MDS layer adds nothing, and a zero block keeps the second half's
// rounds uniform: no select in front of the scalar loads' first use)
static const u64 RC_HOST[372] = {
#include "poseidon_constants.inc"
};
#if defined(__HIPCC__)
static __constant__ u64 RC_DEV[372] = {
#include "poseidon_constants.inc"
};
#endif

#include "poseidon_partial_groups.inc"
static const PartialGroup PG_HOST[7] = POSEIDON_PARTIAL_GROUPS_INIT;
#if defined(__HIPCC__)
static __constant__ PartialGroup PG_DEV[7] = POSEIDON_PARTIAL_GROUPS_INIT;
#endif
GL_HD const PartialGroup& partial_group(int g) {
#if defined(__HIP_DEVICE_COMPILE__)
    return PG_DEV[g];
#else
    return PG_HOST[g];
#endif
}

GL_HD u64 rc(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return RC_DEV[i];
#else
    return RC_HOST[i];
#endif
}

GL_HD u64 sbox(u64 x) {
    const u64 x2 = gl::mul_nc(x, x);
    const u64 x4 = gl::mul_nc(x2, x2);
    const u64 x3 = gl::mul_nc(x2, x);
    return gl::mul_nc(x3, x4);
}

// two S-boxes with their multiplication chains interleaved (hides the dependent-instruction latency of the asm blocks)
GL_HD void sbox2(u64& x, u64& y) {
    u64 x2, y2, x3, y3, x4, y4;
    gl::mul2_nc(x, x, y, y, x2, y2);
    gl::mul2_nc(x2, x2, y2, y2, x4, y4);
    gl::mul2_nc(x2, x, y2, y, x3, y3);
    gl::mul2_nc(x3, x4, y3, y4, x, y);
}

// S-box with x^3 and x^4 computed side by side (three dependent multiplications deep instead of four)
GL_HD u64 sbox_ilp(u64 x) {
    const u64 x2 = gl::mul_nc(x, x);
    u64 x3, x4;
    gl::mul2_nc(x2, x, x2, x2, x3, x4);
    return gl::mul_nc(x3, x4);
}

// acc_lo + acc_hi * 2^32 (both < 2^58) folded to a u64 residue with 2^64 = 2^32 - 1
GL_HD u64 fold96(u64 acc_lo, u64 acc_hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    // acc_lo + (hi_lo + hi_hi 2^32) 2^32 = acc_lo + hi_hi 2^64 + hi_lo 2^32: T = hi_hi (2^32 - 1) + acc_lo as ONE v_mad_u64_u32 (below 2^59:
    // no carry), then hi_lo joins T's high word; that carry selects the single +(2^32 - 1) correction (the wrapped value is below T, so the
    // corrected sum cannot wrap again).  5 VALU (round 5; 7 before: the sum L was formed first, with a carry into H and a move to pair it).
    u32 r0, r1;
    // (scratch through gl.h's names: a translation unit built with GL_ASM_SCRATCH_LOW -- quotient.hip, permutation.hip -- keeps its register
    // budget; with v80 / v81 / v86 spelled out here those kernels were pushed to 87 VGPRs whatever the setting: ADVICE r05)
    asm("v_mad_u64_u32 " GL_P01 ", vcc, %4, -1, %2\n\t"
        "v_add_co_u32_e32 " GL_R1 ", vcc, " GL_R1 ", %3\n\t"
        "v_cndmask_b32_e64 " GL_R6 ", 0, -1, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, " GL_R0 ", " GL_R6 "\n\t"
        "v_addc_co_u32_e64 %1, vcc, " GL_R1 ", 0, vcc"
        : "=&v"(r0), "=&v"(r1)
        : "v"(acc_lo), "v"((u32)acc_hi), "v"((u32)(acc_hi >> 32))
        : GL_R0, GL_R1, GL_R6, "vcc");
    return ((u64)r1 << 32) | r0;
#else
    const u64 L = acc_lo + (acc_hi << 32);
    const u64 H = (acc_hi >> 32) + (L < acc_lo ? 1 : 0);
    const u64 t1 = (H << 32) - H;
    u64 v = L + t1;
    if (v < t1) v += gl::EPS;
    return v;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// One row of the MDS layer as ONE block of 24 multiply-adds: acc_lo = sum x_lo[i] C[i] + (k mod 2^58), acc_hi = sum x_hi[i] C[i] + ((k >> 58) << 26)
// -- together k + sum x[i] C[i] in fold96's lo + hi 2^32 form.  Every coefficient of the circulant is an inline constant (<= 41; row 0's own
// element: 17 + MDS_MATRIX_DIAG[0] = 25), so the coefficients need no registers at all, and the round constant rides in as the 64-bit addend
// of the first multiply-add of each half (split by the scalar unit: no VALU) -- before round 5 it cost a multiply-add by 1 per half (the
// compiler would not take it as an addend: zero-extended SGPR pairs for 24 halves spilled, and a chain started in inline asm is re-associated
// into a separate sum + a 64-bit add).  x: the row's twelve inputs in circulant order (x[0] = the row's own element).
// Bounds for fold96: acc_lo < 2^58 + 2^41, acc_hi < 2^32 + 2^41: T = hi_hi (2^32 - 1) + acc_lo stays far below 2^64.
#define POSEIDON_ROW_OPERANDS                                                                                                              \
    "v"(xl[0]), "v"(xl[1]), "v"(xl[2]), "v"(xl[3]), "v"(xl[4]), "v"(xl[5]), "v"(xl[6]), "v"(xl[7]), "v"(xl[8]), "v"(xl[9]), "v"(xl[10]),   \
    "v"(xl[11]), "v"(xh[0]), "v"(xh[1]), "v"(xh[2]), "v"(xh[3]), "v"(xh[4]), "v"(xh[5]), "v"(xh[6]), "v"(xh[7]), "v"(xh[8]), "v"(xh[9]),  \
    "v"(xh[10]), "v"(xh[11])
template <bool ROW0, bool WITH_K>
__device__ __forceinline__ void mds_row(const u32* xl, const u32* xh, u64 k, u64& acc_lo, u64& acc_hi) {
    if constexpr (WITH_K) {
        if constexpr (ROW0)
            asm("s_mov_b32 s82, %26\n\t"
            "s_and_b32 s83, %27, 0x03ffffff\n\t"
            "s_and_b32 s84, %27, 0xfc000000\n\t"
            "s_mov_b32 s85, 0\n\t"
            "v_mad_u64_u32 %0, vcc, %2, 25, s[82:83]\n\t"
            "v_mad_u64_u32 %1, vcc, %14, 25, s[84:85]\n\t"
            "v_mad_u64_u32 %0, vcc, %3, 15, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %15, 15, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %4, 41, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %16, 41, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %5, 16, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %17, 16, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %6, 2, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %18, 2, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %7, 28, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %19, 28, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %8, 13, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %20, 13, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %9, 13, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %21, 13, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %10, 39, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %22, 39, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %11, 18, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %23, 18, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %12, 34, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %24, 34, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %13, 20, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %25, 20, %1"
                : "=&v"(acc_lo), "=&v"(acc_hi)
                : POSEIDON_ROW_OPERANDS, "s"((u32)k), "s"((u32)(k >> 32))
                : "vcc", "scc", "s82", "s83", "s84", "s85");
        else
            asm("s_mov_b32 s82, %26\n\t"
            "s_and_b32 s83, %27, 0x03ffffff\n\t"
            "s_and_b32 s84, %27, 0xfc000000\n\t"
            "s_mov_b32 s85, 0\n\t"
            "v_mad_u64_u32 %0, vcc, %2, 17, s[82:83]\n\t"
            "v_mad_u64_u32 %1, vcc, %14, 17, s[84:85]\n\t"
            "v_mad_u64_u32 %0, vcc, %3, 15, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %15, 15, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %4, 41, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %16, 41, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %5, 16, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %17, 16, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %6, 2, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %18, 2, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %7, 28, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %19, 28, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %8, 13, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %20, 13, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %9, 13, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %21, 13, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %10, 39, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %22, 39, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %11, 18, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %23, 18, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %12, 34, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %24, 34, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %13, 20, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %25, 20, %1"
                : "=&v"(acc_lo), "=&v"(acc_hi)
                : POSEIDON_ROW_OPERANDS, "s"((u32)k), "s"((u32)(k >> 32))
                : "vcc", "scc", "s82", "s83", "s84", "s85");
    } else {
        if constexpr (ROW0)
            asm("v_mad_u64_u32 %0, vcc, %2, 25, 0\n\t"
            "v_mad_u64_u32 %1, vcc, %14, 25, 0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, 15, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %15, 15, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %4, 41, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %16, 41, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %5, 16, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %17, 16, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %6, 2, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %18, 2, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %7, 28, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %19, 28, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %8, 13, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %20, 13, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %9, 13, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %21, 13, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %10, 39, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %22, 39, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %11, 18, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %23, 18, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %12, 34, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %24, 34, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %13, 20, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %25, 20, %1"
                : "=&v"(acc_lo), "=&v"(acc_hi)
                : POSEIDON_ROW_OPERANDS
                : "vcc");
        else
            asm("v_mad_u64_u32 %0, vcc, %2, 17, 0\n\t"
            "v_mad_u64_u32 %1, vcc, %14, 17, 0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, 15, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %15, 15, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %4, 41, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %16, 41, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %5, 16, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %17, 16, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %6, 2, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %18, 2, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %7, 28, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %19, 28, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %8, 13, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %20, 13, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %9, 13, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %21, 13, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %10, 39, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %22, 39, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %11, 18, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %23, 18, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %12, 34, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %24, 34, %1\n\t"
            "v_mad_u64_u32 %0, vcc, %13, 20, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %25, 20, %1"
                : "=&v"(acc_lo), "=&v"(acc_hi)
                : POSEIDON_ROW_OPERANDS
                : "vcc");
    }
}
#endif

// s <- MDS * s + k (kc: the 12 constants, already in registers; nullptr: none).  s: any u64 residues.
GL_HD void mds_add_const(u64* s, const u64* kc) {
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        lo[i] = (u32)s[i];
        hi[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        u64 acc_lo, acc_hi;
#if defined(__HIP_DEVICE_COMPILE__)
        u32 xl[12], xh[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            xl[i] = lo[(i + r) % 12];
            xh[i] = hi[(i + r) % 12];
        }
        if (kc) {
            if (r == 0) mds_row<true, true>(xl, xh, kc[r], acc_lo, acc_hi);
            else mds_row<false, true>(xl, xh, kc[r], acc_lo, acc_hi);
        } else {
            if (r == 0) mds_row<true, false>(xl, xh, 0, acc_lo, acc_hi);
            else mds_row<false, false>(xl, xh, 0, acc_lo, acc_hi);
        }
#else
        const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        const u64 k = kc ? kc[r] : 0;
        acc_lo = (u32)k, acc_hi = k >> 32;
        for (int i = 0; i < 12; ++i) {
            const u32 c = r == 0 && i == 0 ? C[0] + 8 : C[i];   // row 0: MDS_MATRIX_DIAG[0] = 8 joins the circulant's entry for the same element
            acc_lo += (u64)lo[(i + r) % 12] * c;
            acc_hi += (u64)hi[(i + r) % 12] * c;
        }
#endif
        s[r] = fold96(acc_lo, acc_hi);
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// One row of a fused partial-round group: acc = d3 M1 + d2 m2 + sum x[j] m3[j] + k in fold96's lo + hi 2^32 form, as ONE block of 28
// multiply-adds -- the two halves interleaved, each starting with the constant's share as the 64-bit addend of its first multiply-add (M1, an
// entry of M itself, is an inline constant; the entries of M^2 and M^3 are scalar operands).  The constant is split INSIDE the block on the
// scalar unit (low half: k mod 2^58; high half: (k >> 58) << 26), into the block's own scratch pair: handed in as two ready-made 64-bit
// operands (round 5) the compiler computed all 24 of a group ahead of the S-boxes and spilled them to VGPR lanes (66 SGPR spills in
// leaf_hash_kernel, ~140 v_readlane / v_writelane per permutation: VERDICT r05 weak 3); now the raw constants stay where the scalar loads put them.
template <unsigned M1>
__device__ __forceinline__ void group_row(u32 d3l, u32 d3h, u32 d2l, u32 d2h, u32 m2, u64 k, const u32* xl, const u32* xh, const u32* m3, u64& acc_lo,
                                          u64& acc_hi) {
    static_assert(M1 <= 64, "inline constant");
    asm("s_mov_b32 s82, %43\n\t"
        "s_and_b32 s83, %44, 0x03ffffff\n\t"
        "s_and_b32 s84, %44, 0xfc000000\n\t"
        "s_mov_b32 s85, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %2, %45, s[82:83]\n\t"
        "v_mad_u64_u32 %1, vcc, %3, %45, s[84:85]\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %6, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %5, %6, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %31, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %19, %31, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %32, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %20, %32, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %33, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %21, %33, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %34, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %22, %34, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %35, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %23, %35, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %36, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %24, %36, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %37, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %25, %37, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %38, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %26, %38, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %15, %39, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %27, %39, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %16, %40, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %28, %40, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %17, %41, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %29, %41, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %42, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %30, %42, %1"
        : "=&v"(acc_lo), "=&v"(acc_hi)
        : "v"(d3l), "v"(d3h), "v"(d2l), "v"(d2h), "s"(m2),                                                                           // 2..6
          "v"(xl[0]), "v"(xl[1]), "v"(xl[2]), "v"(xl[3]), "v"(xl[4]), "v"(xl[5]), "v"(xl[6]), "v"(xl[7]), "v"(xl[8]), "v"(xl[9]), "v"(xl[10]),
          "v"(xl[11]),                                                                                                             // 7..18
          "v"(xh[0]), "v"(xh[1]), "v"(xh[2]), "v"(xh[3]), "v"(xh[4]), "v"(xh[5]), "v"(xh[6]), "v"(xh[7]), "v"(xh[8]), "v"(xh[9]), "v"(xh[10]),
          "v"(xh[11]),                                                                                                             // 19..30
          "s"(m3[0]), "s"(m3[1]), "s"(m3[2]), "s"(m3[3]), "s"(m3[4]), "s"(m3[5]), "s"(m3[6]), "s"(m3[7]), "s"(m3[8]), "s"(m3[9]), "s"(m3[10]),
          "s"(m3[11]),                                                                                                             // 31..42
          "s"((u32)k), "s"((u32)(k >> 32)), "n"(M1)                                                                                // 43..45
        : "vcc", "scc", "s82", "s83", "s84", "s85");
}
template <int I> struct GroupRow {
    static __device__ __forceinline__ void run(u64* s, const u64* kv, const u32* lo, const u32* hi, u32 d2l, u32 d2h, u32 d3l, u32 d3h) {
        u32 m3[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) m3[j] = MDS3[I][j];
        u64 acc_lo, acc_hi;
        group_row<MDS1[I][0]>(d3l, d3h, d2l, d2h, MDS2[I][0], kv[I], lo, hi, m3, acc_lo, acc_hi);
        s[I] = fold96(acc_lo, acc_hi);
        if constexpr (I + 1 < 12) GroupRow<I + 1>::run(s, kv, lo, hi, d2l, d2h, d3l, d3h);
    }
};
#endif

// Three consecutive partial rounds in one dense pass (derivation and bounds: tools/gen_poseidon_partial_groups.py).
// The MDS entries are so small that M^2 and M^3 still fit 32-bit multiplicands with room in 64-bit accumulators, so
// instead of 3 x (288 multiply-adds + 12 folds) a group costs 288 (M^3) + 24 + 26 (row 0 of M, M^2) + 48 (the two
// inner S-box corrections) multiply-adds and 14 folds.  s: x1 on entry (round constants included), x1' on exit.
// GATE = true is the PoseidonGate form of the same three rounds (gates.hip): the S-box inputs are the gate's partial_sbox wires
// w[0..3) instead of the computed values, and the computed inputs of rounds 2 and 3 are returned (canonical) in x_out[0..2)
// for the constraints "computed - wire" (round 1's input is s[0] on entry).  GATE = false with x_out != nullptr is the witness
// generator's form: the plain permutation that also reports those two S-box inputs (the gate's wires).
template <bool GATE> GL_HD void partial_group3_core(u64* s, const PartialGroup& G, const u64* w, u64* x_out) {
    // request the group's constants before the first S-box (scalar-load latency hidden under it)
    const u64 k2 = G.k2, k3 = G.k3;
    u64 kv[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) kv[i] = G.kvec[i];
    s[0] = sbox(GATE ? w[0] : s[0]);
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        lo[j] = (u32)s[j];
        hi[j] = (u32)(s[j] >> 32);
    }
    // x2_0 = (M y)[0] + c2[0]
    u64 a_lo = (u32)k2, a_hi = k2 >> 32;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        a_lo += (u64)lo[j] * MDS1[0][j];
        a_hi += (u64)hi[j] * MDS1[0][j];
    }
    // x2, x3 are canonical only where somebody asks for them (the gate's constraints, the generator's wires); the permutation itself runs on
    // residues: S-box, the difference d (gl::sub_a takes any residues) and its 32-bit halves as multiplicands (round 5: 16 instructions
    // per group fewer in the hashing kernels)
    u64 x2 = fold96(a_lo, a_hi);
    if (GATE || x_out) {
        x2 = gl::canon(x2);
        x_out[0] = x2;
    }
    const u64 d2 = gl::sub_a(sbox(GATE ? w[1] : x2), x2);
    const u32 d2l = (u32)d2, d2h = (u32)(d2 >> 32);
    // x3_0 = (M^2 y)[0] + M[0][0] d2 + (M c2)[0] + c3[0]
    u64 b_lo = (u32)k3, b_hi = k3 >> 32;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        b_lo += (u64)lo[j] * MDS2[0][j];
        b_hi += (u64)hi[j] * MDS2[0][j];
    }
    b_lo += (u64)d2l * MDS1[0][0];
    b_hi += (u64)d2h * MDS1[0][0];
    u64 x3 = fold96(b_lo, b_hi);
    if (GATE || x_out) {
        x3 = gl::canon(x3);
        x_out[1] = x3;
    }
    const u64 d3 = gl::sub_a(sbox(GATE ? w[2] : x3), x3);
    const u32 d3l = (u32)d3, d3h = (u32)(d3 >> 32);
    // x1' = M^3 y + d2 (M^2 e0) + d3 (M e0) + kvec
#if defined(__HIP_DEVICE_COMPILE__)
    GroupRow<0>::run(s, kv, lo, hi, d2l, d2h, d3l, d3h);
#else
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const u64 k = kv[i];
        u64 acc_lo = (u32)k, acc_hi = k >> 32;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            acc_lo += (u64)lo[j] * MDS3[i][j];
            acc_hi += (u64)hi[j] * MDS3[i][j];
        }
        acc_lo += (u64)d2l * MDS2[i][0] + (u64)d3l * MDS1[i][0];
        acc_hi += (u64)d2h * MDS2[i][0] + (u64)d3h * MDS1[i][0];
        s[i] = fold96(acc_lo, acc_hi);
    }
#endif
}
template <bool GATE> GL_HD void partial_group3_core(u64* s, int g, const u64* w, u64* x_out) {
    partial_group3_core<GATE>(s, partial_group(g), w, x_out);
}
GL_HD void partial_group3(u64* s, int g) { partial_group3_core<false>(s, g, nullptr, nullptr); }

// in: any u64 residues; out: u64 residues (the caller makes canonical what leaves the sponge -- a chain of absorbs needs that for the digest
// only: 48 instructions per permutation otherwise)
GL_HD void permute_residues(u64* s) {
#if defined(__HIP_DEVICE_COMPILE__)
    // A zero the compiler cannot see through, made anew by every call: the constants of the first round and of round 25 have fixed addresses,
    // so in a sponge loop they are loop-invariant -- the compiler loaded all 48 words once, in front of the loop, found no scalar registers to
    // keep them in and parked them in VGPR lanes: 48 v_readlane per permutation, where a scalar load costs the vector
    // unit nothing.  Indexed by this zero they are loaded where they are used.
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
#else
    constexpr int z = 0;
#endif
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl::add_nc(s[i], rc(z + i));
    // the next round's constants are requested (scalar loads) BEFORE the S-boxes so their latency hides under ~800
    // instructions instead of parking the wave right in front of the MDS layer (17 % of wave cycles in the first version)
    for (int r = 0; r < HALF_FULL; ++r) {
        u64 kc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) kc[i] = rc(12 * (r + 1) + i);
#pragma unroll
        for (int i = 0; i < 12; i += 2) sbox2(s[i], s[i + 1]);
        mds_add_const(s, kc);
    }
    // 22 partial rounds = 7 fused groups of 3 (rounds 4..24) + round 25
    {
        // (the table's address is re-derived by the compiler in front of every group -- one scalar load from the constant pool; making the
        // pointer opaque to stop that turns the group's constants into vector loads: measured, worse)
        for (int g = 0; g < 7; ++g) partial_group3(s, g);
    }
    {
        u64 kc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) kc[i] = rc(z + 12 * (HALF_FULL + N_PARTIAL) + i);
        s[0] = sbox(s[0]);
        mds_add_const(s, kc);
    }
    for (int r = HALF_FULL + N_PARTIAL; r < N_ROUNDS; ++r) {
        u64 kc[12];
        const int next = 12 * (r + 1);                          // r = 29: the zero block behind the constants (the last round adds nothing)
#pragma unroll
        for (int i = 0; i < 12; ++i) kc[i] = rc(next + i);
#pragma unroll
        for (int i = 0; i < 12; i += 2) sbox2(s[i], s[i + 1]);
        mds_add_const(s, kc);
    }
}
// in: any residues (canonical inputs included); out: canonical field elements
GL_HD void permute(u64* s) {
    permute_residues(s);
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl::canon(s[i]);
}

#if defined(__HIPCC__)
// Latency-oriented form: one permutation spread over 16 lanes (lane l < 12 owns state element l; lanes 12..15 idle).
// Used where there are too few independent permutations to fill the chip (upper Merkle levels, FRI trees): the
// lane-per-permutation form has a ~65 us dependent-instruction chain; here the 12 S-boxes of a full round and the 12
// MDS rows run side by side and the state is exchanged through LDS (one ds_write_b64 + 12 ds_read_b64 per round).
// `sh` points at this 16-lane group's private 24-word LDS window (element l is stored at l and l + 12 so the
// circulant index needs no modulo).  All 64 lanes of the wave must call it together.  Returns the canonical value.
constexpr int WIDE_LANES = 16, WIDE_LDS_WORDS = 48;  // 48-word stride keeps the four groups of a wave on disjoint banks
__device__ __forceinline__ u64 permute_wide(u64 x, u64* sh, unsigned l) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const unsigned row = l < 12 ? l : 0;
    // this lane's row of M^3 and its entries of the first columns of M^2 and M (fused partial rounds below)
    u32 m3[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) m3[j] = MDS3[row][j];
    const u32 m2c = MDS2[row][0], m1c = MDS1[row][0];
    x = gl::add_nc(x, rc(row));
    // this lane's round constants are fetched up front (16 independent loads, one memory latency) instead of one exposed
    // lane-indexed load per round on the dependent chain
    u64 kplain[9], kgroup[7];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int r = i < 4 ? i : HALF_FULL + N_PARTIAL - 1 + (i - 4);  // plain rounds: 0..3, 25, 26..29
        kplain[i] = r + 1 < N_ROUNDS ? rc(12 * (r + 1) + row) : 0;
    }
#pragma unroll
    for (int g = 0; g < 7; ++g) kgroup[g] = partial_group(g).kvec[row];
    // one plain round: S-box (full: every lane; partial: lane 0), exchange through LDS, this lane's MDS row + next constants
    auto plain_round = [&](u64 k, bool full) {
        if (full || l == 0) x = sbox_ilp(x);
        // the 16 lanes of a group live in one wave: LDS operations of a wave execute in order, so a wavefront-scope
        // fence (compiler ordering only) is all the synchronisation the exchange needs
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (l < 12) {
            sh[l] = x;
            sh[l + 12] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // three independent accumulator chains per half (latency: 4 dependent multiply-adds instead of 12)
        u64 al[3] = {(u32)k, 0, 0}, ah[3] = {k >> 32, 0, 0};
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const u64 v = sh[row + i];
            al[i % 3] += (u64)(u32)v * C[i];
            ah[i % 3] += (v >> 32) * C[i];
        }
        if (l == 0) {
            al[1] += (u64)(u32)x * 8u;
            ah[1] += (x >> 32) * 8u;
        }
        x = fold96(al[0] + al[1] + al[2], ah[0] + ah[1] + ah[2]);
    };
#pragma unroll
    for (int i = 0; i < HALF_FULL; ++i) plain_round(kplain[i], true);
    // 21 partial rounds as 7 fused groups of three (same algebra as partial_group3_core): ONE exchange per group.  After it every
    // lane holds the whole post-S-box state y, so every lane computes the two inner S-box inputs x2_0 = (M y)[0] + k2 and
    // x3_0 = (M^2 y)[0] + M00 d2 + k3 redundantly (no broadcast), and its own row of M^3 y + d2 M^2[:,0] + d3 M[:,0] + kvec.
    // The M^3 row and most of the x3 dot product do not depend on the inner S-boxes: independent work for the scheduler to place
    // inside their dependent chains.  Critical path per group: 3 S-boxes + 1 LDS round trip instead of 3 + 3.
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        const PartialGroup& G = partial_group(g);
        const u64 k2 = G.k2, k3 = G.k3, kv = kgroup[g];
        if (l == 0) x = sbox_ilp(x);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (l < 12) sh[l] = x;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        u32 lo[12], hi[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const u64 v = sh[j];
            lo[j] = (u32)v;
            hi[j] = (u32)(v >> 32);
        }
        u64 a_lo[2] = {(u32)k2, 0}, a_hi[2] = {k2 >> 32, 0};      // x2_0
        u64 b_lo[2] = {(u32)k3, 0}, b_hi[2] = {k3 >> 32, 0};      // x3_0 without the d2 term
        u64 c_lo[3] = {(u32)kv, 0, 0}, c_hi[3] = {kv >> 32, 0, 0};  // this lane's row of M^3 y + kvec
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            a_lo[j % 2] += (u64)lo[j] * MDS1[0][j];
            a_hi[j % 2] += (u64)hi[j] * MDS1[0][j];
            b_lo[j % 2] += (u64)lo[j] * MDS2[0][j];
            b_hi[j % 2] += (u64)hi[j] * MDS2[0][j];
            c_lo[j % 3] += (u64)lo[j] * m3[j];
            c_hi[j % 3] += (u64)hi[j] * m3[j];
        }
        const u64 x2 = gl::canon(fold96(a_lo[0] + a_lo[1], a_hi[0] + a_hi[1]));
        const u64 d2 = gl::sub(gl::canon(sbox_ilp(x2)), x2);
        const u32 d2l = (u32)d2, d2h = (u32)(d2 >> 32);
        const u64 x3 = gl::canon(fold96(b_lo[0] + b_lo[1] + (u64)d2l * MDS1[0][0], b_hi[0] + b_hi[1] + (u64)d2h * MDS1[0][0]));
        const u64 d3 = gl::sub(gl::canon(sbox_ilp(x3)), x3);
        const u32 d3l = (u32)d3, d3h = (u32)(d3 >> 32);
        x = fold96(c_lo[0] + c_lo[1] + c_lo[2] + (u64)d2l * m2c + (u64)d3l * m1c, c_hi[0] + c_hi[1] + c_hi[2] + (u64)d2h * m2c + (u64)d3h * m1c);
    }
#pragma unroll
    for (int i = 4; i < 9; ++i) plain_round(kplain[i], i > 4);
    return gl::canon(x);
}
#endif

// host-side permutation (Fiat-Shamir transcript: ~125 permutations per step proof between GPU phases).  An AVX2-compiled clone of
// the same code was measured: 35 % faster on the authoring container's Xeon, 13 % SLOWER on the GPU box's EPYC 9575F (1.28 ->
// 1.45 us), so the baseline x86-64 build is what ships.
inline void permute_host(u64* s) { permute(s); }

// ---- host-side sponge helpers (hash/hashing.rs), used by the Challenger and for tiny inputs ----
inline void hash_no_pad_host(const u64* in, size_t n, u64 out[4]) {
    u64 s[12] = {0};
    for (size_t off = 0; off < n; off += 8) {
        const size_t len = n - off < 8 ? n - off : 8;
        for (size_t i = 0; i < len; ++i) s[i] = in[off + i];
        permute_host(s);
    }
    for (int i = 0; i < 4; ++i) out[i] = s[i];
}
}  // namespace poseidon
