// Goldilocks radix-2 NTT / iNTT / coset LDE for gfx950 (replaces plonky2_field 0.2.0 fft.rs `fft_classic`,
// `ifft_with_options`, polynomial/mod.rs `lde` + `coset_fft_with_options`, and plonky2_util `reverse_index_bits`,
// reached from PolynomialBatch::from_values/from_coeffs under prove() -- /root/reference/src/vtfhe/
// ivc_based_vpbs.rs:302,333,364).  Also the reference's own negacyclic NTT
// (/root/reference/src/vtfhe/crypto/poly.rs:9-64) as a batched kernel.
//
// Design (MI355X): decimation-in-frequency, natural order in, bit-reversed order out -- which IS plonky2's leaf order,
// so the transpose + reverse_index_bits passes of the reference disappear.  The rate-8 LDE is computed as 8
// independent size-n coset transforms of coeff_i * (7 w^r)^i (never a zero-padded size-8n transform).
// A transform is two launches: a strided pass (stages 0..log_r-1, tile = R rows x C adjacent columns in LDS, 128-B
// row segments coalesced) and a contiguous pass (remaining stages on 2048-element tiles, in place).  Twiddles are
// read from an HBM table of w^j (L2-resident: 128 KiB at n = 2^15).
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

// Register radix-8 rounds over an LDS tile of TILE elements (THREADS x 8).  A round performs up to three consecutive
// DIF stages on the 8 values a thread holds, so the tile crosses LDS once per 3 stages instead of once per stage.
// Stage "bits": the butterfly distance of a stage is 2^bit in tile-index space.  In a round with stage bits
// b1 > b2 > b3 the thread owns the 8 tile indices that differ only in those bits; rounds with fewer than three stages
// left fill the spare positions with unused low/high bits (those values just ride along).
// tw(idx_lo, stage) returns the twiddle-table index of the butterfly whose low element sits at tile index idx_lo.
// LDS swizzle: physical slot = idx ^ ((idx >> 3) & 31).  It is GF(2)-linear and a bijection on every aligned group of
// 32 elements, and it makes all three lane patterns of the radix-8 rounds conflict-free for ds_read/write_b64
// (lanes varying index bits {0..4}, {0,1,2,6,7} or {3..7}: see DESIGN.md) -- the unswizzled tile had 4- and 8-way
// bank conflicts in the second and third round.
__device__ __forceinline__ unsigned sw(unsigned idx) { return idx ^ ((idx >> 3) & 31u); }

__device__ __forceinline__ unsigned insert_zero_bit(unsigned x, unsigned pos) {
    return ((x >> pos) << (pos + 1)) | (x & ((1u << pos) - 1));
}

// A full round (three stages) is evaluated as a true radix-8 butterfly: the three radix-2 twiddles of element k factor as
// W1 w_8^k, W2 w_4^(k&1), W4 with W1 = w^e1 the twiddle of the round's first butterfly, W2 = W1^2, W4 = W1^4 -- so the
// 8th roots are applied inside (w_8 = -2^24, w_4 = 2^48, w_8^3 = -2^72: shifts, gl::mul_2e*), and each output gets ONE table
// twiddle w^(j e1), j = 1..7: 7 modular multiplications + 5 shifts per 8 elements instead of 12 multiplications.
// `inverse`: the table holds powers of w^-1, whose 8th roots are the conjugates (w_8^-1 = 2^72, w_4^-1 = -2^48, w_8^-3 = 2^24).
// Where a round's twiddles come from.  Gathering w^(j e1) from the table of all n powers costs a 64-line memory instruction per twiddle in
// the rounds whose exponents differ from lane to lane (the SQ counters of round 1 showed these kernels waiting, not computing: VALU issue
// share 0.12).  Which twiddles a thread needs depends only on (transform size, pass, tile, round, thread) -- not on the column or the
// coset -- so they are laid out once per transform size in exactly the order the threads consume them: rt[(round slot)][thread], a
// coalesced 512-byte load per wave and twiddle, shared by every column and coset through L2.  RECORD: the pass that writes that layout
// (run once per transform size on a dummy tile, by the same code that later reads it).
template <bool RECORD> struct TwSource {
    const u64* roots;   // w^i, i < n
    u64* rt;            // this block's round table (slots x THREADS)
    unsigned slot;
    __device__ __forceinline__ u64 get(unsigned exponent) {
        u64* p = rt + (size_t)slot * THREADS + threadIdx.x;
        ++slot;
        if (RECORD) {
            const u64 v = roots[exponent];
            *p = v;
            return v;
        }
        return *p;
    }
};
// twiddle slots a pass of n_stages stages uses per thread: 7 per full round, 4 per stage of a partial one
__host__ __device__ inline unsigned round_slots(unsigned n_stages) { return 7 * (n_stages / 3) + 4 * (n_stages % 3); }

template <bool RECORD, typename TwIndex>
__device__ __forceinline__ void dif_rounds(u64* tile, unsigned n_stages, unsigned first_bit, TwSource<RECORD> tw, bool inverse,
                                           TwIndex tw_index) {
    // stage j (0-based inside this pass) has distance bit first_bit - j
    for (unsigned j0 = 0; j0 < n_stages; j0 += 3) {
        const unsigned ns = n_stages - j0 < 3 ? n_stages - j0 : 3;
        // active bits, descending; spare bits chosen below the lowest active bit or above the highest
        unsigned b[3];
        b[0] = first_bit - j0;
        b[1] = ns > 1 ? b[0] - 1 : (b[0] >= 1 ? b[0] - 1 : b[0] + 1);
        b[2] = ns > 2 ? b[0] - 2 : (b[0] >= 2 ? b[0] - 2 : b[0] + (ns > 1 ? 1 : 2));
        // sort descending so that zero-bit insertion goes from the lowest position up
        unsigned p0 = b[0], p1 = b[1], p2 = b[2];
        if (p0 < p1) { unsigned t = p0; p0 = p1; p1 = t; }
        if (p1 < p2) { unsigned t = p1; p1 = p2; p2 = t; }
        if (p0 < p1) { unsigned t = p0; p0 = p1; p1 = t; }
        unsigned base = insert_zero_bit(insert_zero_bit(insert_zero_bit(threadIdx.x, p2), p1), p0);
        const unsigned pbase = sw(base);  // sw is linear: sw(base | kbits) = sw(base) ^ sw(kbits), kbits wave-uniform
        // element k: bit b[0] <- k>>2, b[1] <- (k>>1)&1, b[2] <- k&1   (b[] in stage order, not sorted order)
        u64 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = tile[pbase ^ sw((((k >> 2) & 1u) << b[0]) | (((k >> 1) & 1u) << b[1]) | ((k & 1u) << b[2]))];
        if (ns == 3) {
            const unsigned e1 = tw_index(base, j0);  // exponent of W1; < n/8 because the three active bits of `base` are zero
            u64 a[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = gl::add(x[k], x[k + 4]);
            const u64 d0 = gl::sub(x[0], x[4]);
            a[4] = d0;
            if (!inverse) {
                a[5] = gl::mul_2e24(gl::sub(x[5], x[1]));  // (x1 - x5) w_8,   w_8 = -2^24
                a[6] = gl::mul_2e48(gl::sub(x[2], x[6]));  // (x2 - x6) w_8^2
                a[7] = gl::mul_2e72(gl::sub(x[7], x[3]));  // (x3 - x7) w_8^3, w_8^3 = -2^72
            } else {
                a[5] = gl::mul_2e72(gl::sub(x[1], x[5]));
                a[6] = gl::mul_2e48(gl::sub(x[6], x[2]));
                a[7] = gl::mul_2e24(gl::sub(x[3], x[7]));
            }
            u64 bq[8];
#pragma unroll
            for (int h = 0; h < 8; h += 4) {
                bq[h] = gl::add(a[h], a[h + 2]);
                bq[h + 2] = gl::sub(a[h], a[h + 2]);
                bq[h + 1] = gl::add(a[h + 1], a[h + 3]);
                bq[h + 3] = gl::mul_2e48(inverse ? gl::sub(a[h + 3], a[h + 1]) : gl::sub(a[h + 1], a[h + 3]));  // times w_4
            }
            // outputs: element k carries W_j with j = bit-reversal of k over 3 bits
            // the seven twiddles are requested together, ahead of the butterfly arithmetic that precedes their use
            const u64 t4 = tw.get(4 * e1), t2 = tw.get(2 * e1), t6 = tw.get(6 * e1), t1 = tw.get(e1), t5 = tw.get(5 * e1), t3 = tw.get(3 * e1),
                      t7 = tw.get(7 * e1);
            x[0] = gl::add(bq[0], bq[1]);
            x[1] = gl::mul(gl::sub(bq[0], bq[1]), t4);
            x[2] = gl::mul(gl::add(bq[2], bq[3]), t2);
            x[3] = gl::mul(gl::sub(bq[2], bq[3]), t6);
            x[4] = gl::mul(gl::add(bq[4], bq[5]), t1);
            x[5] = gl::mul(gl::sub(bq[4], bq[5]), t5);
            x[6] = gl::mul(gl::add(bq[6], bq[7]), t3);
            x[7] = gl::mul(gl::sub(bq[6], bq[7]), t7);
        } else {
        // stage A: pairs (k, k+4)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned lo = base | (((k >> 1) & 1u) << b[1]) | ((k & 1u) << b[2]);
                const u64 u = x[k], v = x[k + 4];
                x[k] = gl::add(u, v);
                x[k + 4] = gl::mul(gl::sub(u, v), tw.get(tw_index(lo, j0)));
            }
        }
        if (ns > 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k & 2) continue;  // pairs (k, k+2)
                const unsigned lo = base | (((k >> 2) & 1u) << b[0]) | ((k & 1u) << b[2]);
                const u64 u = x[k], v = x[k + 2];
                x[k] = gl::add(u, v);
                x[k + 2] = gl::mul(gl::sub(u, v), tw.get(tw_index(lo, j0 + 1)));
            }
        }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[pbase ^ sw((((k >> 2) & 1u) << b[0]) | (((k >> 1) & 1u) << b[1]) | ((k & 1u) << b[2]))] = x[k];
        __syncthreads();
    }
}

// Strided pass: stages [0, log_r).  Tile = R rows x C cols, element (rho, gamma) <-> index rho*(n/R) + c0 + gamma.
template <bool RECORD>
__global__ void __launch_bounds__(THREADS)
ntt_strided_kernel(const u64* __restrict__ in, u64* __restrict__ out, const u64* __restrict__ prescale,
                   const u64* __restrict__ roots, u64* __restrict__ round_tables, unsigned log_n, unsigned log_r, size_t in_col_stride,
                   size_t out_col_stride, unsigned rate_bits, unsigned block_first, int inverse) {
    __shared__ u64 tile[TILE];
    // this tile's table: the strided pass's exponents depend on the tile's columns
    TwSource<RECORD> tw{roots, round_tables + (size_t)blockIdx.x * round_slots(log_r) * THREADS, 0};
    const unsigned log_c = TILE_LOG - log_r;
    const unsigned C = 1u << log_c, R = 1u << log_r;
    const unsigned row_stride = 1u << (log_n - log_r);  // n / R
    const unsigned c0 = blockIdx.x << log_c;
    // blockIdx.z = local leaf block; leaf block B holds coset brev(B)
    const unsigned coset = gl::bitrev32(block_first + blockIdx.z, rate_bits);
    const u64* src = in + blockIdx.y * in_col_stride;
    const u64* ps = prescale ? prescale + ((size_t)coset << log_n) : nullptr;
    for (unsigned t = threadIdx.x; t < TILE; t += THREADS) {
        const unsigned rho = t >> log_c, gamma = t & (C - 1);
        const unsigned idx = rho * row_stride + c0 + gamma;
        u64 x = RECORD ? 0 : src[idx];
        if (!RECORD && ps) x = gl::mul(x, ps[idx]);
        tile[sw(t)] = x;
    }
    __syncthreads();
    // stage s: butterfly distance (R >> (s+1)) rows = tile bit log_c + log_r - 1 - s
    dif_rounds(tile, log_r, log_c + log_r - 1, tw, inverse != 0, [=](unsigned lo, unsigned s) {
        const unsigned half_rows = R >> (s + 1);
        const unsigned rho = lo >> log_c, gamma = lo & (C - 1);
        return ((rho & (half_rows - 1)) * row_stride + c0 + gamma) << s;
    });
    if (RECORD) return;
    u64* dst = out + blockIdx.y * out_col_stride + ((size_t)blockIdx.z << log_n);
    for (unsigned t = threadIdx.x; t < TILE; t += THREADS) {
        const unsigned rho = t >> log_c, gamma = t & (C - 1);
        dst[rho * row_stride + c0 + gamma] = tile[sw(t)];
    }
}

// Contiguous pass: stages [s_begin, log_n) on blocks of B = n >> s_begin elements; one workgroup owns
// min(TILE, n) consecutive elements.  If s_begin == 0 the input is read from `in` (with optional prescale), else the
// transform continues in place in `out`.  bitrev_out: scatter to natural order and scale (inverse transform).
template <bool RECORD>
__global__ void __launch_bounds__(THREADS)
ntt_contig_kernel(const u64* __restrict__ in, u64* __restrict__ out, const u64* __restrict__ prescale,
                  const u64* __restrict__ roots, u64* __restrict__ round_tables, unsigned log_n, unsigned s_begin, size_t in_col_stride,
                  size_t out_col_stride, unsigned rate_bits, int bitrev_out, u64 scale, unsigned block_first) {
    __shared__ u64 tile[TILE];
    // one table for every tile: ((base + lo) & (half - 1)) << s does not depend on the tile (base is a multiple of 2 half)
    TwSource<RECORD> tw{roots, round_tables, 0};
    const unsigned n = 1u << log_n;
    const unsigned tile_elems = n < TILE ? n : TILE;
    const unsigned base = blockIdx.x * tile_elems;
    const unsigned coset = gl::bitrev32(block_first + blockIdx.z, rate_bits);
    const size_t coset_off = (size_t)blockIdx.z << log_n;
    u64* dst_col = out + blockIdx.y * out_col_stride;
    if (RECORD) {
        for (unsigned t = threadIdx.x; t < tile_elems; t += THREADS) tile[sw(t)] = 0;
    } else if (s_begin == 0) {
        const u64* src = in + blockIdx.y * in_col_stride;
        const u64* ps = prescale ? prescale + ((size_t)coset << log_n) : nullptr;
        for (unsigned t = threadIdx.x; t < tile_elems; t += THREADS) {
            u64 x = src[base + t];
            if (ps) x = gl::mul(x, ps[base + t]);
            tile[sw(t)] = x;
        }
    } else {
        // in place continuation; for the inverse transform the strided pass wrote to `in` (scratch)
        const u64* src = bitrev_out ? in + blockIdx.y * in_col_stride : dst_col + coset_off;
        for (unsigned t = threadIdx.x; t < tile_elems; t += THREADS) tile[sw(t)] = src[base + t];
    }
    __syncthreads();
    if (tile_elems == TILE) {
        // stage s: distance n >> (s+1) = tile bit log_n - 1 - s
        dif_rounds(tile, log_n - s_begin, log_n - 1 - s_begin, tw, bitrev_out != 0, [=](unsigned lo, unsigned j) {
            const unsigned s = s_begin + j;
            const unsigned half = n >> (s + 1);
            return ((base + lo) & (half - 1)) << s;
        });
    } else {
        // small transforms (n < 2048): plain radix-2 sweeps
        for (unsigned s = s_begin; s < log_n; ++s) {
            const unsigned half = n >> (s + 1);
            for (unsigned k = threadIdx.x; k < tile_elems / 2; k += THREADS) {
                const unsigned lo = ((k / half) * 2 * half) + (k % half);
                const u64 u = tile[sw(lo)], v = tile[sw(lo + half)];
                tile[sw(lo)] = gl::add(u, v);
                tile[sw(lo + half)] = gl::mul(gl::sub(u, v), roots[((base + lo) & (half - 1)) << s]);
            }
            __syncthreads();
        }
    }
    if (RECORD) return;
    if (bitrev_out) {
        for (unsigned t = threadIdx.x; t < tile_elems; t += THREADS)
            dst_col[gl::bitrev32(base + t, log_n)] = gl::mul(tile[sw(t)], scale);
    } else {
        for (unsigned t = threadIdx.x; t < tile_elems; t += THREADS) dst_col[coset_off + base + t] = tile[sw(t)];
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

// Two passes cover log_n <= 2 * TILE_LOG: the strided pass does log_r stages, the contiguous pass the remaining log_n - log_r, which
// must fit one tile.  log_r = 7 keeps 16-element (128 B) runs in the strided pass; above 2^18 points it has to grow (2^19: the
// quotient's inverse transform at degree 2^16 -- the first version kept 7 there and left 12 stages to an 11-stage tile).
unsigned split_log_r(unsigned log_n) {
    if (log_n <= TILE_LOG) return 0;
    const unsigned want = log_n - 9 > 7 ? 7 : log_n - 9, need = log_n - TILE_LOG;
    return want > need ? want : need;
}
// round tables of a transform size, stored behind the n powers of the root table: [contiguous pass | strided pass, tile by tile]
struct RoundLayout {
    unsigned log_r, tiles;
    size_t contig_words, strided_words;
    bool tables;   // the tile-sized passes use them (transforms of at least one tile)
};
RoundLayout round_layout(unsigned log_n) {
    RoundLayout l{};
    l.log_r = split_log_r(log_n);
    const size_t n = (size_t)1 << log_n;
    l.tiles = n <= TILE ? 1 : (unsigned)(n / TILE);
    l.tables = n >= TILE;
    l.contig_words = l.tables ? (size_t)round_slots(log_n - l.log_r) * THREADS : 0;
    l.strided_words = l.log_r ? (size_t)l.tiles * round_slots(l.log_r) * THREADS : 0;
    return l;
}
}  // namespace

size_t root_table_words(unsigned log_n) {
    const RoundLayout l = round_layout(log_n);
    return ((size_t)1 << log_n) + l.contig_words + l.strided_words;
}

void launch_root_table(hipStream_t s, u64* roots, unsigned log_n, bool inverse) {
    u64 w = gl::root_of_unity(log_n);
    if (inverse) w = gl::inv(w);
    const size_t cnt = (size_t)1 << log_n;
    hipLaunchKernelGGL(root_table_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, roots, log_n, w);
    // the round tables: the transform kernels themselves in RECORD mode, one column, one coset, on a dummy tile
    const RoundLayout l = round_layout(log_n);
    if (!l.tables) return;
    u64* contig = roots + cnt;
    u64* strided = contig + l.contig_words;
    if (l.log_r)
        hipLaunchKernelGGL(ntt_strided_kernel<true>, dim3(l.tiles, 1, 1), dim3(THREADS), 0, s, (const u64*)nullptr, (u64*)nullptr, (const u64*)nullptr,
                           (const u64*)roots, strided, log_n, l.log_r, (size_t)0, (size_t)0, 0u, 0u, inverse ? 1 : 0);
    hipLaunchKernelGGL(ntt_contig_kernel<true>, dim3(1, 1, 1), dim3(THREADS), 0, s, (const u64*)nullptr, (u64*)nullptr, (const u64*)nullptr,
                       (const u64*)roots, contig, log_n, l.log_r, (size_t)0, (size_t)0, 0u, inverse ? 1 : 0, (u64)1, 0u);
}

void launch_prescale_table(hipStream_t s, u64* table, unsigned log_n, unsigned rate_bits, u64 shift) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(prescale_table_kernel, dim3((n + 255) / 256, 1u << rate_bits), dim3(256), 0, s, table, log_n, rate_bits,
                       shift, gl::root_of_unity(log_n + rate_bits));
}

static void run_transform(hipStream_t s, const u64* in, u64* out, u64* scratch, const u64* prescale, const u64* roots,
                          unsigned ncols, unsigned log_n, unsigned rate_bits, bool inverse, size_t in_stride, size_t out_stride,
                          unsigned block_first, unsigned n_blocks) {
    const unsigned n = 1u << log_n;
    const unsigned cosets = n_blocks;
    if (log_n > 2 * TILE_LOG) throw DeviceError{VPBS_ERR_INVALID, "transform larger than 2^22 points"};
    const RoundLayout l = round_layout(log_n);
    const unsigned log_r = l.log_r;
    const u64 scale = inverse ? gl::inv((u64)n) : 1;
    const unsigned tiles = l.tiles;
    // the round tables sit behind the powers (root_table_words); the table of an inverse transform was recorded with the inverse roots
    u64* contig_rt = const_cast<u64*>(roots) + n;
    u64* strided_rt = contig_rt + l.contig_words;
    if (log_r == 0) {
        hipLaunchKernelGGL(ntt_contig_kernel<false>, dim3(tiles, ncols, cosets), dim3(THREADS), 0, s, in, out, prescale, roots, contig_rt, log_n, 0u,
                           in_stride, out_stride, rate_bits, inverse ? 1 : 0, scale, block_first);
        return;
    }
    if (inverse) {
        // strided pass into scratch (layout [ncols][n]), contiguous pass scatters into `out`
        hipLaunchKernelGGL(ntt_strided_kernel<false>, dim3(tiles, ncols, 1), dim3(THREADS), 0, s, in, scratch, (const u64*)nullptr, roots, strided_rt,
                           log_n, log_r, in_stride, (size_t)n, 0u, 0u, 1);
        hipLaunchKernelGGL(ntt_contig_kernel<false>, dim3(tiles, ncols, 1), dim3(THREADS), 0, s, (const u64*)scratch, out,
                           (const u64*)nullptr, roots, contig_rt, log_n, log_r, (size_t)n, out_stride, 0u, 1, scale, 0u);
    } else {
        hipLaunchKernelGGL(ntt_strided_kernel<false>, dim3(tiles, ncols, cosets), dim3(THREADS), 0, s, in, out, prescale, roots, strided_rt, log_n,
                           log_r, in_stride, out_stride, rate_bits, block_first, 0);
        hipLaunchKernelGGL(ntt_contig_kernel<false>, dim3(tiles, ncols, cosets), dim3(THREADS), 0, s, (const u64*)nullptr, out,
                           (const u64*)nullptr, roots, contig_rt, log_n, log_r, (size_t)0, out_stride, rate_bits, 0, scale, block_first);
    }
}

void launch_intt(hipStream_t s, const u64* values, u64* coeffs, u64* scratch, const u64* inv_roots, unsigned ncols, unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    run_transform(s, values, coeffs, scratch, nullptr, inv_roots, ncols, log_n, 0, true, n, n, 0, 1);
}

void launch_coset_lde(hipStream_t s, const u64* coeffs, u64* out, const u64* roots, const u64* prescale, unsigned ncols,
                      unsigned log_n, unsigned rate_bits, unsigned block_first, unsigned n_blocks) {
    const size_t n = (size_t)1 << log_n;
    if (n_blocks == 0) n_blocks = 1u << rate_bits;
    run_transform(s, coeffs, out, nullptr, prescale, roots, ncols, log_n, rate_bits, false, n, n * n_blocks, block_first, n_blocks);
}

void launch_negacyclic(hipStream_t s, u64* data, const u64* table, unsigned batch, unsigned log_n, bool inverse, u64 ninv) {
    hipLaunchKernelGGL(negacyclic_kernel, dim3(batch), dim3(THREADS), 0, s, data, table, log_n, inverse ? 1 : 0, ninv);
}
}  // namespace vpbs
