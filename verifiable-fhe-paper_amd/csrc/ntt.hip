// Goldilocks radix-2 NTT / iNTT / coset LDE for gfx950 (replaces plonky2_field 0.2.0 fft.rs `fft_classic`,
// `ifft_with_options`, polynomial/mod.rs `lde` + `coset_fft_with_options`, and plonky2_util `reverse_index_bits`,
// reached from PolynomialBatch::from_values/from_coeffs under prove() -- /root/reference/src/vtfhe/
// ivc_based_vpbs.rs:302,333,364).  Also the reference's own negacyclic NTT
// (/root/reference/src/vtfhe/crypto/poly.rs:9-64) as a batched kernel.
//
// Design (MI355X): decimation-in-frequency, natural order in, bit-reversed order out -- which IS plonky2's leaf order,
// so the transpose + reverse_index_bits passes of the reference disappear.  The rate-8 LDE is computed as 8
// independent size-n coset transforms of coeff_i * (7 w^r)^i (never a zero-padded size-8n transform).
// From 2^12 points on a transform is two launches of radix-16 register rounds (a strided pass and a contiguous pass over 4096-element
// tiles; below); smaller transforms run in one workgroup.
#include <cstdlib>

#define GL_ASM_SCRATCH_LOW 1  // these kernels need ~40 VGPRs of their own: keep the asm scratch block low (occupancy)
#include "context.h"
#include "kernels.h"

namespace vpbs {
namespace {
constexpr unsigned TILE_LOG = 11;
constexpr unsigned TILE = 1u << TILE_LOG;  // 2048 elements = 16 KiB of LDS
constexpr unsigned THREADS = 256;

__global__ void root_table_kernel(u64* roots, unsigned log_n, u64 w) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < ((size_t)1 << log_n)) roots[i] = gl::pow(w, i);
}

__global__ void prescale_table_kernel(u64* table, unsigned log_n, unsigned rate_bits, u64 shift, u64 w_big) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const unsigned r = blockIdx.y;
    if (i < ((size_t)1 << log_n)) table[((size_t)r << log_n) + i] = gl::pow(gl::mul(shift, gl::pow(w_big, r)), i);
}

// Transforms below 2^12 points (FRI's last rounds, small test circuits): ONE workgroup per column and coset, the whole transform in LDS as
// radix-2 decimation-in-frequency sweeps, natural order in, bit-reversed order out (forward) or natural order out + 1/n (inverse).  Not a
// hot path: every transform of the prover's commitments has 2^12 points or more and takes the radix-16 passes below.
__global__ void __launch_bounds__(THREADS)
ntt_small_kernel(const u64* __restrict__ in, u64* __restrict__ out, const u64* __restrict__ prescale, const u64* __restrict__ roots,
                 unsigned log_n, size_t in_col_stride, size_t out_col_stride, unsigned rate_bits, int inverse, u64 scale, unsigned block_first) {
    __shared__ u64 tile[TILE];
    const unsigned n = 1u << log_n;
    const unsigned coset = gl::bitrev32(block_first + blockIdx.z, rate_bits);   // leaf block B holds coset brev(B)
    const u64* src = in + blockIdx.y * in_col_stride;
    const u64* ps = prescale ? prescale + ((size_t)coset << log_n) : nullptr;
    for (unsigned t = threadIdx.x; t < n; t += THREADS) {
        u64 x = src[t];
        if (ps) x = gl::mul(x, ps[t]);
        tile[t] = x;
    }
    __syncthreads();
    for (unsigned s = 0; s < log_n; ++s) {
        const unsigned half = n >> (s + 1);
        for (unsigned k = threadIdx.x; k < n / 2; k += THREADS) {
            const unsigned lo = ((k / half) * 2 * half) + (k % half);
            const u64 u = tile[lo], v = tile[lo + half];
            tile[lo] = gl::add(u, v);
            tile[lo + half] = gl::mul(gl::sub(u, v), roots[(lo & (half - 1)) << s]);
        }
        __syncthreads();
    }
    u64* dst_col = out + blockIdx.y * out_col_stride;
    if (inverse) {
        for (unsigned t = threadIdx.x; t < n; t += THREADS) dst_col[gl::bitrev32(t, log_n)] = gl::mul(tile[t], scale);
    } else {
        u64* dst = dst_col + ((size_t)blockIdx.z << log_n);
        for (unsigned t = threadIdx.x; t < n; t += THREADS) dst[t] = tile[t];
    }
}

// Negacyclic transform of the reference, one polynomial per workgroup (n <= 2048 per LDS tile), in place.
// forward (poly.rs:9-34): for m = 1,2,..: t = n/2m; (u, v*S[m+i]) -> (u+v, u-v)
// backward (poly.rs:36-64): for m = n/2,..,1: (u, v) -> (u+v, (u-v)*S[m+i]); then * N^-1
__global__ void __launch_bounds__(THREADS)
negacyclic_kernel(u64* __restrict__ data, const u64* __restrict__ table, unsigned log_n, int inverse, u64 ninv) {
    __shared__ u64 tile[TILE];
    const unsigned n = 1u << log_n;
    u64* p = data + (size_t)blockIdx.x * n;
    for (unsigned t = threadIdx.x; t < n; t += THREADS) tile[t] = p[t];
    __syncthreads();
    if (!inverse) {
        for (unsigned m = 1; m < n; m <<= 1) {
            const unsigned t = n / (2 * m);
            for (unsigned k = threadIdx.x; k < n / 2; k += THREADS) {
                const unsigned i = k / t, j = 2 * i * t + (k % t);
                const u64 u = tile[j], v = gl::mul(tile[j + t], table[m + i]);
                tile[j] = gl::add(u, v);
                tile[j + t] = gl::sub(u, v);
            }
            __syncthreads();
        }
        for (unsigned t = threadIdx.x; t < n; t += THREADS) p[t] = tile[t];
    } else {
        for (unsigned m = n >> 1; m >= 1; m >>= 1) {
            const unsigned t = n / (2 * m);
            for (unsigned k = threadIdx.x; k < n / 2; k += THREADS) {
                const unsigned i = k / t, j = 2 * i * t + (k % t);
                const u64 u = tile[j], v = tile[j + t];
                tile[j] = gl::add(u, v);
                tile[j + t] = gl::mul(gl::sub(u, v), table[m + i]);
            }
            __syncthreads();
        }
        for (unsigned t = threadIdx.x; t < n; t += THREADS) p[t] = gl::mul(tile[t], ninv);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Transforms of 2^12 points and more: radix-16 register rounds, decimation in time with BLOCK twiddles.
//
// A coset transform y[pos] = sum_i x_i (s w^brev(pos))^i (s = 1: the plain transform; w^-1 and a factor 1/n: the inverse) is computed
// natural order in, bit-reversed order out by Cooley-Tukey stages whose twiddle depends on the block only: stage t, block b (the index
// bits above the butterfly's distance bit) multiplies the upper input by T(t, b) = (s w^brev_t(b))^(n / 2^(t+1)).  The coset shift is
// part of T -- there is no prescale pass and no prescale table (one modular multiplication per point less than scaling the
// coefficients first).  q consecutive stages on the 2^q elements that differ in the q active index bits collapse into
//     z_k = x_k tau^k  (k < 2^q),  tau = (s w^brev_t(b))^(n / 2^(t+q)),     followed by the PURE DFT of size 2^q on z,
// and the pure DFT's own twiddles are 16th roots of unity, which in Goldilocks are powers of two up to sign (2 has order 192: plonky2's
// w_16 = 2^156 = -2^60, w_16^2 = w_8 = -2^24, w_16^3 = 2^84, w_4 = 2^48, w_16^5 = 2^12, w_16^6 = -2^72, w_16^7 = -2^36): shifts.
// So a radix-16 round costs 15 table multiplications + 17 shifts + 64 additions per 16 points per FOUR stages, and a thread that holds
// 16 values crosses LDS once per four stages.  (fft.rs semantics: the result equals fft_classic + reverse_index_bits)
//
// Two passes per transform (a tile = 4096 elements, 256 threads x 16 values, 32 KiB of LDS):
//   strided pass     stages [0, A): tile = 2^A rows x 2^(12-A) adjacent columns; global -> registers -> ... -> registers -> global
//   contiguous pass  stages [A, L): tile = 4096 consecutive elements; the last round leaves 16 consecutive elements in a thread, which a
//                    final trip through LDS turns into coalesced stores
// A pass of S stages = ceil(S / 4) rounds: radix-16 rounds, then one round of radix 2^QL (QL = S - 4 (rounds - 1)) in which the thread's
// 16 values are 2^(4-QL) independent butterflies.  2^16 points = [4 4 | 4 4]: two LDS crossings + the output transposition where the
// radix-8 form (3 3 1 | 3 3 3) made eight; 2^17 = [4 4 | 4 4 1]; 2^19 = [4 4 | 4 4 3]; 2^20 = [4 4 | 4 4 4].
// Twiddles: per (coset, round) a table tau^k laid out [k][block]: a wave's loads are contiguous (or one address for all its lanes).
constexpr unsigned T16_LOG = 12, T16 = 1u << T16_LOG, T16_THREADS = 256;

// LDS swizzle of the 4096-element tile (GF(2)-linear, a bijection).  A thread of a round owns the 16 indices base | (j << o); the lanes of
// a ds_write_b64 group (16 lanes) / ds_read_b64 group (32 lanes) differ in the lowest 4 / 5 index bits outside [o, o + 4).  For every
// o in 0..8, for the linear order of the forward transform's final read and for the bit-reversed order of the inverse transform's (lanes
// differ in bits 11..7) those lanes land on distinct bank pairs: bits 7..4 are folded onto bits 3..0, bit 8 onto bit 4 and bits 11..9
// onto bits 2..0 (worked through in DESIGN.md 4).
__device__ __forceinline__ constexpr unsigned swz(unsigned idx) {
    return idx ^ ((idx >> 4) & 15u) ^ (((idx >> 8) & 1u) << 4) ^ ((idx >> 9) & 7u);
}

// Inside a transform values are ARBITRARY u64 residues (gl::add_a / sub_a / mul_nc take and return any residue): no product, shift or sum
// pays for a canonical form; the contiguous pass -- always the last -- canonicalises what it stores.
// x 2^E mod p for 0 < E < 96 (any u64 in, a u64 residue out): the forms of gl::mul_2e24 / 48 / 72 for every multiple of 12
template <unsigned E>
__device__ __forceinline__ u64 mul_2e(u64 x) {
    static_assert(E > 0 && E < 96 && E != 32 && E != 64, "shift out of range");
    if constexpr (E < 32) {
        // x 2^E = lo + hi 2^64, hi < 2^E
        return gl::reduce96_asm(x << E, (u32)(x >> (64 - E)));
    } else if constexpr (E < 64) {
        // x 2^E = lo + h 2^64, h = h1 2^32 + h0 < 2^E: the 128-bit reduction of a product (2^64 = 2^32 - 1, 2^96 = -1)
        const u64 h = x >> (64 - E);
        return gl::reduce128_asm(x << E, (u32)h, (u32)(h >> 32));
    } else {
        constexpr unsigned F = E - 64;               // x 2^F = c 2^64 + b 2^32 + a  ->  (x 2^F) 2^64 = a (2^32 - 1) - b - c 2^32
        const u64 a = (x << F) & gl::EPS, b = (x >> (32 - F)) & gl::EPS, c = x >> (64 - F);
        const u64 pos = (a << 32) - a;               // < p
        return gl::sub(pos, b + (c << 32));          // both canonical: the plain form
    }
}

// pure DFTs, natural order in, bit-reversed order out, in place; INV: with the inverse roots
template <bool INV>
__device__ __forceinline__ void dft4(u64& x0, u64& x1, u64& x2, u64& x3) {
    const u64 b0 = gl::add_a(x0, x2), b2 = gl::sub_a(x0, x2), b1 = gl::add_a(x1, x3);
    const u64 b3 = mul_2e<48>(INV ? gl::sub_a(x3, x1) : gl::sub_a(x1, x3));   // times w_4 = 2^48 (inverse: -2^48)
    x0 = gl::add_a(b0, b1);
    x1 = gl::sub_a(b0, b1);
    x2 = gl::add_a(b2, b3);
    x3 = gl::sub_a(b2, b3);
}
template <bool INV>
__device__ __forceinline__ void dft8(u64* x) {
    u64 a[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = gl::add_a(x[k], x[k + 4]);
    a[4] = gl::sub_a(x[0], x[4]);
    if (!INV) {
        a[5] = mul_2e<24>(gl::sub_a(x[5], x[1]));   // (x1 - x5) w_8,   w_8 = -2^24
        a[6] = mul_2e<48>(gl::sub_a(x[2], x[6]));   // w_8^2 = 2^48
        a[7] = mul_2e<72>(gl::sub_a(x[7], x[3]));   // w_8^3 = -2^72
    } else {
        a[5] = mul_2e<72>(gl::sub_a(x[1], x[5]));   // w_8^-1 = 2^72
        a[6] = mul_2e<48>(gl::sub_a(x[6], x[2]));   // w_8^-2 = -2^48
        a[7] = mul_2e<24>(gl::sub_a(x[3], x[7]));   // w_8^-3 = 2^24
    }
    dft4<INV>(a[0], a[1], a[2], a[3]);
    dft4<INV>(a[4], a[5], a[6], a[7]);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = a[k];
}
template <bool INV>
__device__ __forceinline__ void dft16(u64* z) {
    u64 a[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = gl::add_a(z[k], z[k + 8]);
    a[8] = gl::sub_a(z[0], z[8]);
    if (!INV) {
        a[9] = mul_2e<60>(gl::sub_a(z[9], z[1]));     // w_16   = -2^60
        a[10] = mul_2e<24>(gl::sub_a(z[10], z[2]));   // w_16^2 = -2^24
        a[11] = mul_2e<84>(gl::sub_a(z[3], z[11]));   // w_16^3 =  2^84
        a[12] = mul_2e<48>(gl::sub_a(z[4], z[12]));   // w_16^4 =  2^48
        a[13] = mul_2e<12>(gl::sub_a(z[5], z[13]));   // w_16^5 =  2^12
        a[14] = mul_2e<72>(gl::sub_a(z[14], z[6]));   // w_16^6 = -2^72
        a[15] = mul_2e<36>(gl::sub_a(z[15], z[7]));   // w_16^7 = -2^36
    } else {
        a[9] = mul_2e<36>(gl::sub_a(z[1], z[9]));     // w_16^-1 =  2^36
        a[10] = mul_2e<72>(gl::sub_a(z[2], z[10]));   // w_16^-2 =  2^72
        a[11] = mul_2e<12>(gl::sub_a(z[11], z[3]));   // w_16^-3 = -2^12
        a[12] = mul_2e<48>(gl::sub_a(z[12], z[4]));   // w_16^-4 = -2^48
        a[13] = mul_2e<84>(gl::sub_a(z[13], z[5]));   // w_16^-5 = -2^84
        a[14] = mul_2e<24>(gl::sub_a(z[6], z[14]));   // w_16^-6 =  2^24
        a[15] = mul_2e<60>(gl::sub_a(z[7], z[15]));   // w_16^-7 =  2^60
    }
    dft8<INV>(a);
    dft8<INV>(a + 8);
#pragma unroll
    for (int k = 0; k < 16; ++k) z[k] = a[k];
}

// One register round of radix 2^Q on the thread's 16 values: x[(sub << Q) | k] is element k of butterfly `sub` (the 4 - Q owned bits above the
// active ones select the butterfly and are the low bits of its block index).  tw: this round's table, row k = tau^k over `nb` blocks; SCALE0:
// row 0 holds a factor for element 0 as well (1/n in the first round of an inverse transform), otherwise row 0 is not read.
template <unsigned Q, bool SCALE0>
__device__ __forceinline__ void round16_twiddles(u64 (&t)[16], const u64* __restrict__ tw, size_t nb, unsigned b_thread) {
    constexpr unsigned M = 1u << Q, SUBS = 16u >> Q;
#pragma unroll
    for (unsigned sub = 0; sub < SUBS; ++sub) {
        const size_t b = (size_t)b_thread * SUBS + sub;
#pragma unroll
        for (unsigned k = SCALE0 ? 0 : 1; k < M; ++k) t[sub * M + k] = tw[(size_t)k * nb + b];
    }
}
// the round itself on twiddles that are already on their way (a pass requests the next round's before it crosses LDS: the loads then fly
// while the tile is exchanged instead of standing in front of the round's first products)
template <unsigned Q, bool INV, bool SCALE0>
__device__ __forceinline__ void round16(u64 (&x)[16], const u64 (&t)[16]) {
    constexpr unsigned M = 1u << Q, SUBS = 16u >> Q;
#pragma unroll
    for (unsigned sub = 0; sub < SUBS; ++sub) {
#pragma unroll
        for (unsigned k = SCALE0 ? 0 : 1; k < M; ++k) x[sub * M + k] = gl::mul_nc(x[sub * M + k], t[sub * M + k]);
        u64* z = x + sub * M;
        if constexpr (Q == 4) dft16<INV>(z);
        else if constexpr (Q == 3) dft8<INV>(z);
        else if constexpr (Q == 2) dft4<INV>(z[0], z[1], z[2], z[3]);
        else {
            const u64 u = z[0], v = z[1];
            z[0] = gl::add_a(u, v);
            z[1] = gl::sub_a(u, v);
        }
    }
}

// tile index of element 0 of the thread whose 16 elements are base | (j << o)
__device__ __forceinline__ unsigned owner_base(unsigned tid, unsigned o) { return (tid & ((1u << o) - 1u)) | ((tid >> o) << (o + 4)); }

struct Pass16 {
    const u64* tables;      // [coset][words_per_coset]
    size_t coset_words;
    size_t round_off[3];    // this pass's rounds inside a coset's tables
    unsigned round_t[3];    // global stage at which each round starts (its table has 2^t blocks)
};

// One pass of S = 4 (NR - 1) + QL stages.  STRIDED: stages [0, S) of the transform, rows = the top S index bits, a tile = all 2^S rows x
// 2^(12-S) adjacent columns.  Otherwise: the last S stages, a tile = 4096 consecutive elements.  INV: inverse roots; the first round of the
// first pass of an inverse transform also carries the factor 1/n (SCALE0).  The contiguous pass of an inverse transform writes natural order.  in_block_stride: elements between the inputs of two leaf blocks (0: all cosets read one coefficient column; n: the
// contiguous pass continuing in place).
template <unsigned NR, unsigned QL, bool STRIDED, bool INV, bool SCALE0>
__global__ void __launch_bounds__(T16_THREADS)
ntt16_kernel(const u64* in, u64* out /* may be `in`: the contiguous pass of a forward transform continues in place */, Pass16 P, unsigned log_n, size_t in_col_stride, size_t in_block_stride,
             size_t out_col_stride, unsigned rate_bits, unsigned block_first) {
    constexpr unsigned S = 4 * (NR - 1) + QL;
    static_assert(NR >= 2 && NR <= 3 && QL >= 1 && QL <= 4 && S <= 12, "pass shape");
    static_assert(!STRIDED || S <= 10, "a strided tile keeps at least four adjacent columns");
    constexpr unsigned LOG_C = T16_LOG - S;                 // strided: adjacent columns of a tile
    constexpr unsigned TOP = STRIDED ? T16_LOG - 1 : S - 1; // tile bit of the pass's first stage
    constexpr unsigned O_LAST = STRIDED ? LOG_C : 0;        // lowest owned bit in the last round
    __shared__ u64 tile[T16];
    const unsigned tid = threadIdx.x;
    const unsigned coset = gl::bitrev32(block_first + blockIdx.z, rate_bits);   // leaf block B holds coset brev(B)
    const u64* tw = P.tables + (size_t)coset * P.coset_words;
    const size_t n = (size_t)1 << log_n;
    const size_t row_stride = n >> (STRIDED ? S : 0);       // strided: elements between two rows of the tile
    // The contiguous pass of an INVERSE transform writes natural order, position brev_L(g) for the element at g: its tile is not 4096
    // consecutive elements but the 2^(12-S) blocks of 2^S elements whose block numbers differ in their TOP bits -- after the bit reversal
    // those are the lowest address bits, so the tile's results form runs of 2^(12-S) consecutive elements (128 bytes at 2^16 points)
    // instead of 4096 single words 2^(L-12) elements apart.
    constexpr bool BITREV = INV && !STRIDED;
    const size_t tile_base = STRIDED ? ((size_t)blockIdx.x << LOG_C) : (BITREV ? ((size_t)blockIdx.x << S) : ((size_t)blockIdx.x << T16_LOG));
    // global element of tile index idx
    auto gidx = [&](unsigned idx) -> size_t {
        if constexpr (STRIDED) return (size_t)(idx >> LOG_C) * row_stride + tile_base + (idx & ((1u << LOG_C) - 1u));
        else if constexpr (BITREV && S < T16_LOG) return ((size_t)(idx >> S) << (log_n - T16_LOG + S)) | tile_base | (idx & ((1u << S) - 1u));
        else return tile_base + idx;
    };
    const u64* src = in + blockIdx.y * in_col_stride + blockIdx.z * in_block_stride;   // in_block_stride 0: every coset reads the same coefficients
    u64 x[16], t[16];
    constexpr unsigned O0 = TOP - 3, O1 = NR == 3 ? TOP - 7 : O_LAST;
    auto block_of = [&](unsigned base, unsigned o) -> unsigned {   // the index bits above the thread's butterflies: their block
        return (unsigned)(STRIDED ? (base >> (o + 4)) : (gidx(base) >> (o + 4)));
    };
    // round 0: straight from global memory (a wave's lanes cover the adjacent columns / elements: 128-byte runs at least)
    {
        constexpr unsigned o = O0;
        const unsigned base = owner_base(tid, o);
        const size_t g0 = gidx(base);
        const size_t step = STRIDED ? (row_stride << (o - LOG_C)) : ((size_t)1 << o);
        round16_twiddles<4, SCALE0>(t, tw + P.round_off[0], (size_t)1 << P.round_t[0], block_of(base, o));
#pragma unroll
        for (unsigned j = 0; j < 16; ++j) x[j] = src[g0 + j * step];
        round16<4, INV, SCALE0>(x, t);
        // the next round's twiddles are requested before the tile crosses LDS
        if constexpr (NR == 3) round16_twiddles<4, false>(t, tw + P.round_off[1], (size_t)1 << P.round_t[1], block_of(owner_base(tid, O1), O1));
        else round16_twiddles<QL, false>(t, tw + P.round_off[1], (size_t)1 << P.round_t[1], block_of(owner_base(tid, O1), O1));
#pragma unroll
        for (unsigned j = 0; j < 16; ++j) tile[swz(base) ^ swz(j << o)] = x[j];
    }
    __syncthreads();
    if constexpr (NR == 3) {
        constexpr unsigned o = O1;
        const unsigned base = owner_base(tid, o);
#pragma unroll
        for (unsigned j = 0; j < 16; ++j) x[j] = tile[swz(base) ^ swz(j << o)];
        __syncthreads();
        round16<4, INV, false>(x, t);
        round16_twiddles<QL, false>(t, tw + P.round_off[2], (size_t)1 << P.round_t[2], block_of(owner_base(tid, O_LAST), O_LAST));
#pragma unroll
        for (unsigned j = 0; j < 16; ++j) tile[swz(base) ^ swz(j << o)] = x[j];
        __syncthreads();
    }
    // last round
    {
        constexpr unsigned o = O_LAST;
        const unsigned base = owner_base(tid, o);
#pragma unroll
        for (unsigned j = 0; j < 16; ++j) x[j] = tile[swz(base) ^ swz(j << o)];
        round16<QL, INV, false>(x, t);
        if constexpr (STRIDED) {
            // rows base_row + j: the lanes of a wave still cover the adjacent columns
            u64* dst = out + blockIdx.y * out_col_stride + ((size_t)blockIdx.z << log_n);
            const size_t g0 = gidx(base);
#pragma unroll
            for (unsigned j = 0; j < 16; ++j) dst[g0 + j * row_stride] = x[j];
        } else {
            __syncthreads();
#pragma unroll
            for (unsigned j = 0; j < 16; ++j) tile[swz(base) ^ swz(j)] = gl::canon(x[j]);   // the transform's results leave in canonical form
            __syncthreads();
            u64* dst_col = out + blockIdx.y * out_col_stride;
            if constexpr (BITREV) {
                // e = (brev_S(u) << (12-S)) | brev(block-in-tile): consecutive e = consecutive addresses inside a run; the tile index it names is brev_12(e)
                const size_t run_base = (size_t)gl::bitrev32(blockIdx.x, log_n - T16_LOG) << (T16_LOG - S);
#pragma unroll
                for (unsigned j = 0; j < 16; ++j) {
                    const unsigned e = j * T16_THREADS + tid;
                    const unsigned ti = __brev(e) >> (32 - T16_LOG);
                    dst_col[((size_t)(e >> (T16_LOG - S)) << (log_n - S)) | run_base | (e & ((1u << (T16_LOG - S)) - 1u))] = tile[swz(ti)];
                }
            } else {
                u64* dst = dst_col + ((size_t)blockIdx.z << log_n) + tile_base;
#pragma unroll
                for (unsigned j = 0; j < 16; ++j) {
                    const unsigned ti = j * T16_THREADS + tid;
                    dst[ti] = tile[swz(ti)];
                }
            }
        }
    }
}

// The shape of a transform of 2^L points, L >= 12: A strided stages + B contiguous stages, each pass radix-16 rounds and a last round of
// radix 2^QL.  2^12: one contiguous pass.
struct Plan16 {
    unsigned L, A, B;
    unsigned s_nr, s_ql, c_nr, c_ql;   // rounds / last radix of the strided and of the contiguous pass (s_nr = 0: no strided pass)
    unsigned n_rounds;
    unsigned t[6], q[6];               // global start stage and radix of every round, strided pass first
    size_t off[6], coset_words;        // table of round r at off[r] (2^q rows x 2^t blocks)
};
Plan16 plan16(unsigned L) {
    Plan16 p{};
    p.L = L;
    p.B = L <= 12 ? L : (L <= 16 ? 8 : (L <= 20 ? L - 8 : 12));
    p.A = L - p.B;
    auto shape = [](unsigned S, unsigned& nr, unsigned& ql) {
        nr = (S + 3) / 4;
        ql = S - 4 * (nr - 1);
    };
    if (p.A) shape(p.A, p.s_nr, p.s_ql);
    shape(p.B, p.c_nr, p.c_ql);
    unsigned r = 0, t = 0;
    size_t off = 0;
    auto add = [&](unsigned nr, unsigned ql) {
        for (unsigned i = 0; i < nr; ++i) {
            const unsigned q = i + 1 < nr ? 4 : ql;
            p.t[r] = t;
            p.q[r] = q;
            p.off[r] = off;
            off += (size_t)1 << (t + q);
            t += q;
            ++r;
        }
    };
    if (p.A) add(p.s_nr, p.s_ql);
    add(p.c_nr, p.c_ql);
    p.n_rounds = r;
    p.coset_words = off;
    return p;
}
bool uses_radix16(unsigned log_n) { return log_n >= T16_LOG; }

struct Plan16Dev {
    unsigned n_rounds, t[6], q[6];
    size_t off[6], coset_words;
};
// table[coset][round][k][b] = scale(round 0 only) * ((shift w_big^coset) w^brev_t(b))^(k n / 2^(t+q));  w = w_n (or its inverse), w_big the
// primitive (n 2^rate_bits)-th root
__global__ void table16_kernel(u64* __restrict__ table, Plan16Dev pl, unsigned log_n, u64 w, u64 shift, u64 w_big, u64 scale) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= pl.coset_words) return;
    const unsigned coset = blockIdx.y;
    unsigned r = 0;
    while (r + 1 < pl.n_rounds && i >= pl.off[r + 1]) ++r;
    const unsigned t = pl.t[r], q = pl.q[r];
    const size_t local = i - pl.off[r];
    const unsigned k = (unsigned)(local >> t), b = (unsigned)(local & (((size_t)1 << t) - 1));
    const u64 s = gl::mul(shift, gl::pow(w_big, coset));
    const u64 base = gl::mul(s, gl::pow(w, gl::bitrev32(b, t)));
    const u64 tau = gl::pow(base, (u64)1 << (log_n - t - q));
    u64 v = gl::pow(tau, k);
    if (r == 0) v = gl::mul(v, scale);
    table[(size_t)coset * pl.coset_words + i] = v;
}
void launch_table16(hipStream_t s, u64* table, unsigned log_n, unsigned n_cosets_log, u64 shift, bool inverse) {
    const Plan16 p = plan16(log_n);
    Plan16Dev d{};
    d.n_rounds = p.n_rounds;
    d.coset_words = p.coset_words;
    for (unsigned r = 0; r < p.n_rounds; ++r) {
        d.t[r] = p.t[r];
        d.q[r] = p.q[r];
        d.off[r] = p.off[r];
    }
    u64 w = gl::root_of_unity(log_n);
    if (inverse) w = gl::inv(w);
    const u64 scale = inverse ? gl::inv((u64)1 << log_n) : 1;
    hipLaunchKernelGGL(table16_kernel, dim3((unsigned)((p.coset_words + 255) / 256), 1u << n_cosets_log), dim3(256), 0, s, table, d, log_n, w, shift,
                       gl::root_of_unity(log_n + n_cosets_log), scale);
}

template <bool STRIDED, bool INV, bool SCALE0>
void launch_pass16(hipStream_t s, unsigned nr, unsigned ql, dim3 grid, const u64* in, u64* out, const Pass16& P, unsigned log_n, size_t in_stride,
                   size_t in_block_stride, size_t out_stride, unsigned rate_bits, unsigned block_first) {
#define VPBS_PASS16(NR_, QL_)                                                                                                              \
    if (nr == NR_ && ql == QL_) {                                                                                                          \
        hipLaunchKernelGGL((ntt16_kernel<NR_, QL_, STRIDED, INV, SCALE0>), grid, dim3(T16_THREADS), 0, s, in, out, P, log_n, in_stride,     \
                           in_block_stride, out_stride, rate_bits, block_first);                                                           \
        return;                                                                                                                            \
    }
    VPBS_PASS16(2, 1) VPBS_PASS16(2, 2) VPBS_PASS16(2, 3) VPBS_PASS16(2, 4)
    if constexpr (STRIDED) {
        VPBS_PASS16(3, 1) VPBS_PASS16(3, 2)
    } else {
        VPBS_PASS16(3, 1) VPBS_PASS16(3, 2) VPBS_PASS16(3, 3) VPBS_PASS16(3, 4)
    }
#undef VPBS_PASS16
    throw DeviceError{VPBS_ERR_INVALID, "no radix-16 pass of that shape"};
}

// tables: plan16(log_n).coset_words words per coset (launch_table16)
template <bool INV>
void run_transform16(hipStream_t s, const u64* in, u64* out, u64* scratch, const u64* tables, unsigned ncols, unsigned log_n, unsigned rate_bits,
                     size_t in_stride, size_t out_stride, unsigned block_first, unsigned n_blocks) {
    const Plan16 p = plan16(log_n);
    const size_t n = (size_t)1 << log_n;
    Pass16 sp{tables, p.coset_words, {0, 0, 0}, {0, 0, 0}}, cp = sp;
    const unsigned s_rounds = p.A ? p.s_nr : 0;
    for (unsigned r = 0; r < s_rounds; ++r) {
        sp.round_off[r] = p.off[r];
        sp.round_t[r] = p.t[r];
    }
    for (unsigned r = 0; r < p.c_nr; ++r) {
        cp.round_off[r] = p.off[s_rounds + r];
        cp.round_t[r] = p.t[s_rounds + r];
    }
    const unsigned tiles = (unsigned)(n >> T16_LOG);
    if (!p.A) {
        launch_pass16<false, INV, INV>(s, p.c_nr, p.c_ql, dim3(tiles, ncols, n_blocks), in, out, cp, log_n, in_stride, (size_t)0, out_stride, rate_bits,
                                       block_first);
        return;
    }
    if (INV) {
        // strided pass into scratch ([ncols][n]), contiguous pass from there to natural order in `out`
        launch_pass16<true, true, true>(s, p.s_nr, p.s_ql, dim3(tiles, ncols, 1), in, scratch, sp, log_n, in_stride, (size_t)0, n, 0u, 0u);
        launch_pass16<false, true, false>(s, p.c_nr, p.c_ql, dim3(tiles, ncols, 1), (const u64*)scratch, out, cp, log_n, n, (size_t)0, out_stride, 0u, 0u);
    } else {
        launch_pass16<true, false, false>(s, p.s_nr, p.s_ql, dim3(tiles, ncols, n_blocks), in, out, sp, log_n, in_stride, (size_t)0, out_stride,
                                          rate_bits, block_first);
        launch_pass16<false, false, false>(s, p.c_nr, p.c_ql, dim3(tiles, ncols, n_blocks), (const u64*)out, out, cp, log_n, out_stride, n, out_stride,
                                           rate_bits, block_first);
    }
}
}  // namespace

// ---- tables ----
// The n powers of the root; from 2^12 points on the radix-16 passes' tables for the plain transform of that direction follow them.
size_t root_table_words(unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    return n + (uses_radix16(log_n) ? plan16(log_n).coset_words : 0);
}

void launch_root_table(hipStream_t s, u64* roots, unsigned log_n, bool inverse) {
    u64 w = gl::root_of_unity(log_n);
    if (inverse) w = gl::inv(w);
    const size_t cnt = (size_t)1 << log_n;
    hipLaunchKernelGGL(root_table_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, roots, log_n, w);
    if (uses_radix16(log_n)) launch_table16(s, roots + cnt, log_n, 0, 1, inverse);   // shift 1, one coset, 1/n for the inverse
}

void launch_prescale_table(hipStream_t s, u64* table, unsigned log_n, unsigned rate_bits, u64 shift) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(prescale_table_kernel, dim3((n + 255) / 256, 1u << rate_bits), dim3(256), 0, s, table, log_n, rate_bits,
                       shift, gl::root_of_unity(log_n + rate_bits));
}

size_t lde_table_words(unsigned log_n, unsigned rate_bits) {
    return uses_radix16(log_n) ? plan16(log_n).coset_words << rate_bits : (size_t)1 << (log_n + rate_bits);
}
void launch_lde_table(hipStream_t s, u64* table, unsigned log_n, unsigned rate_bits, u64 shift) {
    if (uses_radix16(log_n)) launch_table16(s, table, log_n, rate_bits, shift, false);
    else launch_prescale_table(s, table, log_n, rate_bits, shift);
}

void launch_intt(hipStream_t s, const u64* values, u64* coeffs, u64* scratch, const u64* inv_roots, unsigned ncols, unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    if (log_n > 22) throw DeviceError{VPBS_ERR_INVALID, "transform larger than 2^22 points"};
    if (uses_radix16(log_n)) {
        run_transform16<true>(s, values, coeffs, scratch, inv_roots + n, ncols, log_n, 0, n, n, 0, 1);
        return;
    }
    hipLaunchKernelGGL(ntt_small_kernel, dim3(1, ncols, 1), dim3(THREADS), 0, s, values, coeffs, (const u64*)nullptr, inv_roots, log_n, n, n, 0u, 1,
                       gl::inv((u64)n), 0u);
}

void launch_coset_lde(hipStream_t s, const u64* coeffs, u64* out, const u64* roots, const u64* lde_table, unsigned ncols,
                      unsigned log_n, unsigned rate_bits, unsigned block_first, unsigned n_blocks) {
    const size_t n = (size_t)1 << log_n;
    if (log_n > 22) throw DeviceError{VPBS_ERR_INVALID, "transform larger than 2^22 points"};
    if (n_blocks == 0) n_blocks = 1u << rate_bits;
    if (uses_radix16(log_n)) {
        run_transform16<false>(s, coeffs, out, nullptr, lde_table, ncols, log_n, rate_bits, n, n * n_blocks, block_first, n_blocks);
        return;
    }
    hipLaunchKernelGGL(ntt_small_kernel, dim3(1, ncols, n_blocks), dim3(THREADS), 0, s, coeffs, out, lde_table, roots, log_n, n, n * n_blocks, rate_bits,
                       0, (u64)1, block_first);
}

void launch_negacyclic(hipStream_t s, u64* data, const u64* table, unsigned batch, unsigned log_n, bool inverse, u64 ninv) {
    hipLaunchKernelGGL(negacyclic_kernel, dim3(batch), dim3(THREADS), 0, s, data, table, log_n, inverse ? 1 : 0, ninv);
}
}  // namespace vpbs
