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

static const u64 RC_HOST[360] = {
#include "poseidon_constants.inc"
};
#if defined(__HIPCC__)
static __constant__ u64 RC_DEV[360] = {
#include "poseidon_constants.inc"
};
#endif

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

// s <- MDS * s + k, where k = rc[k_off .. k_off+12) (k_off < 0: no constant).  s: any u64 residues.
GL_HD void mds_add_const(u64* s, int k_off) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        lo[i] = (u32)s[i];
        hi[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        u64 k = k_off >= 0 ? rc(k_off + r) : 0;
        u64 acc_lo = (u32)k, acc_hi = k >> 32;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            acc_lo += (u64)lo[(i + r) % 12] * C[i];
            acc_hi += (u64)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) {  // MDS_MATRIX_DIAG[0] = 8
            acc_lo += (u64)lo[0] * 8u;
            acc_hi += (u64)hi[0] * 8u;
        }
        // value = acc_lo + acc_hi * 2^32  (< 2^76): fold the part above 2^64 with 2^64 = 2^32 - 1
        const u64 L = acc_lo + (acc_hi << 32);
        const u64 H = (acc_hi >> 32) + (L < acc_lo ? 1 : 0);
        const u64 t1 = (H << 32) - H;
        u64 v = L + t1;
        if (v < t1) v += gl::EPS;
        s[r] = v;
    }
}

// in/out: canonical field elements
GL_HD void permute(u64* s) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl::add_nc(s[i], rc(i));
    for (int r = 0; r < HALF_FULL; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
        mds_add_const(s, 12 * (r + 1));
    }
    for (int r = HALF_FULL; r < HALF_FULL + N_PARTIAL; ++r) {
        s[0] = sbox(s[0]);
        mds_add_const(s, 12 * (r + 1));
    }
    for (int r = HALF_FULL + N_PARTIAL; r < N_ROUNDS; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
        mds_add_const(s, r + 1 < N_ROUNDS ? 12 * (r + 1) : -1);
    }
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
__device__ __forceinline__ u64 permute_wide(u64 x, volatile u64* sh, unsigned l) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const unsigned row = l < 12 ? l : 0;
    x = gl::add_nc(x, rc(row));
    for (int r = 0; r < N_ROUNDS; ++r) {
        const bool full = r < HALF_FULL || r >= HALF_FULL + N_PARTIAL;
        if (full || l == 0) x = sbox(x);
        __builtin_amdgcn_wave_barrier();
        if (l < 12) {
            sh[l] = x;
            sh[l + 12] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u64 k = r + 1 < N_ROUNDS ? rc(12 * (r + 1) + row) : 0;
        u64 acc_lo = (u32)k, acc_hi = k >> 32;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const u64 v = sh[row + i];
            acc_lo += (u64)(u32)v * C[i];
            acc_hi += (v >> 32) * C[i];
        }
        if (l == 0) {
            acc_lo += (u64)(u32)x * 8u;
            acc_hi += (x >> 32) * 8u;
        }
        const u64 L = acc_lo + (acc_hi << 32);
        const u64 H = (acc_hi >> 32) + (L < acc_lo ? 1 : 0);
        const u64 t1 = (H << 32) - H;
        u64 v = L + t1;
        if (v < t1) v += gl::EPS;
        x = v;
        __builtin_amdgcn_wave_barrier();
    }
    return gl::canon(x);
}
#endif

// ---- host-side sponge helpers (hash/hashing.rs), used by the Challenger and for tiny inputs ----
inline void hash_no_pad_host(const u64* in, size_t n, u64 out[4]) {
    u64 s[12] = {0};
    for (size_t off = 0; off < n; off += 8) {
        const size_t len = n - off < 8 ? n - off : 8;
        for (size_t i = 0; i < len; ++i) s[i] = in[off + i];
        permute(s);
    }
    for (int i = 0; i < 4; ++i) out[i] = s[i];
}
}  // namespace poseidon
