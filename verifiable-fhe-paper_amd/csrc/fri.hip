// Opening evaluation, alpha-combination, division by (X - z), FRI folding and query gathering for gfx950.
// Replaces plonky2 0.2.0 plonk/proof.rs `OpeningSet::new` (p.to_extension().eval(z)), util/reducing.rs
// `ReducingFactor::{reduce_polys_base, shift_poly}`, field polynomial `divide_by_linear`, fri/prover.rs
// `fri_committed_trees` (reduce_with_powers fold) and `fri_prover_query_rounds` (tree.get / tree.prove gathers) --
// reached from prove() at /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364.
// All of these are streaming passes over coefficient columns (HBM-bound).  Field arithmetic is exact, so the parallel
// forms below (dot products with a power table, suffix-scan division) give the same canonical values as the
// reference's sequential Horner loops.
#define GL_ASM_SCRATCH_LOW 1  // low asm scratch block: these kernels need few registers of their own (occupancy)
#include "kernels.h"

namespace vpbs {
namespace {
constexpr unsigned THREADS = 256;

__device__ __forceinline__ gl::Ext load_ext(const u64* p, size_t i) {
    const ulonglong2 v = reinterpret_cast<const ulonglong2*>(p)[i];
    return gl::Ext{v.x, v.y};
}
__device__ __forceinline__ void store_ext(u64* p, size_t i, gl::Ext e) {
    reinterpret_cast<ulonglong2*>(p)[i] = make_ulonglong2(e.c0, e.c1);
}

struct PowerPoints {
    gl::Ext z[4], stride[4];   // stride[t] = z_t^(n / per)
};
// grid (ceil(n / per / 256), count): table t (at out + 2 n t) holds z_t^i, i < n.  A thread raises z to ITS index once (square and multiply) and
// walks on by z^(n / per): `per` entries for one exponentiation + per - 1 multiplications (an exponentiation per entry was 2.5 k instructions,
// 0.018 G wave-instructions per step proof); the walk's stores stay coalesced (entry i + k n / per of thread i).
__global__ void ext_powers_kernel(PowerPoints pts, size_t n, unsigned per, u64* out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, lanes = n / per;
    if (i >= lanes) return;
    u64* table = out + 2 * n * blockIdx.y;
    const gl::Ext stride = pts.stride[blockIdx.y];
    gl::Ext cur = gl::pow(pts.z[blockIdx.y], i);
    for (unsigned k = 0; k < per; ++k) {
        store_ext(table, i + k * lanes, cur);
        cur = gl::mul(cur, stride);
    }
}

// block-wide sum of extension elements (blockDim.x == THREADS); result valid in thread 0
__device__ gl::Ext block_sum(gl::Ext v, u64* sh /* [2*THREADS] */) {
    sh[threadIdx.x] = v.c0;
    sh[THREADS + threadIdx.x] = v.c1;
    __syncthreads();
    for (unsigned d = THREADS / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            sh[threadIdx.x] = gl::add(sh[threadIdx.x], sh[threadIdx.x + d]);
            sh[THREADS + threadIdx.x] = gl::add(sh[THREADS + threadIdx.x], sh[THREADS + threadIdx.x + d]);
        }
        __syncthreads();
    }
    return gl::Ext{sh[0], sh[THREADS]};
}

// grid (chunks, ncols): partial[c][chunk] = sum over the chunk of coeffs[c][i] * zpow[i]
__global__ void __launch_bounds__(THREADS)
eval_partial_kernel(const u64* __restrict__ coeffs, size_t n, size_t col_stride, const u64* __restrict__ zpow,
                    unsigned chunk_len, u64* __restrict__ partial) {
    __shared__ u64 sh[2 * THREADS];
    const u64* col = coeffs + blockIdx.y * col_stride;
    const size_t begin = (size_t)blockIdx.x * chunk_len;
    gl::LazyAcc a0{}, a1{};   // sum of coefficient x power, reduced once per thread (16 instead of ~45 instructions per term)
    for (size_t i = begin + threadIdx.x; i < begin + chunk_len && i < n; i += THREADS) {
        const gl::Ext z = load_ext(zpow, i);
        const u64 v = col[i];
        a0.mac_v(v, z.c0);
        a1.mac_v(v, z.c1);
    }
    const gl::Ext tot = block_sum(gl::Ext{gl::canon(a0.reduce()), gl::canon(a1.reduce())}, sh);
    if (threadIdx.x == 0) store_ext(partial, (size_t)blockIdx.y * gridDim.x + blockIdx.x, tot);
}
__global__ void eval_finish_kernel(const u64* __restrict__ partial, unsigned chunks, unsigned ncols, u64* __restrict__ out) {
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    gl::Ext acc = gl::ext(0);
    for (unsigned k = 0; k < chunks; ++k) acc = gl::add(acc, load_ext(partial, (size_t)c * chunks + k));
    store_ext(out, c, acc);
}

// the same for several coefficient matrices in one launch (the openings of a step proof: four oracles at zeta + the Z columns
// at g zeta): blockIdx.y = global column over all segments
__global__ void __launch_bounds__(THREADS)
eval_partial_multi_kernel(EvalSegments segs, size_t n, unsigned chunk_len, u64* __restrict__ partial) {
    __shared__ u64 sh[2 * THREADS];
    unsigned c = blockIdx.y, k = 0;
    while (k + 1 < segs.count && c >= segs.seg[k].ncols) c -= segs.seg[k++].ncols;
    const u64* col = segs.seg[k].coeffs + (size_t)c * segs.seg[k].col_stride;
    const u64* zpow = segs.seg[k].zpow;
    const size_t begin = (size_t)blockIdx.x * chunk_len;
    gl::LazyAcc a0{}, a1{};   // sum of coefficient x power, reduced once per thread (16 instead of ~45 instructions per term)
    for (size_t i = begin + threadIdx.x; i < begin + chunk_len && i < n; i += THREADS) {
        const gl::Ext z = load_ext(zpow, i);
        const u64 v = col[i];
        a0.mac_v(v, z.c0);
        a1.mac_v(v, z.c1);
    }
    const gl::Ext tot = block_sum(gl::Ext{gl::canon(a0.reduce()), gl::canon(a1.reduce())}, sh);
    if (threadIdx.x == 0) store_ext(partial, (size_t)blockIdx.y * gridDim.x + blockIdx.x, tot);
}

// grid (n / 256, groups): each workgroup row sums a slice of the polynomials (n = 2^15 coefficients alone would leave the
// chip at half a wave per SIMD); the slices are added by combine_finish_kernel.
__global__ void __launch_bounds__(THREADS)
combine_kernel(const u64* const* __restrict__ polys, unsigned n_polys, const u64* __restrict__ alpha_pows, size_t n,
               u64* __restrict__ part0, u64* __restrict__ part1) {
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (i >= n) return;
    const unsigned per = (n_polys + gridDim.y - 1) / gridDim.y;
    const unsigned j0 = blockIdx.y * per, j1 = j0 + per < n_polys ? j0 + per : n_polys;
    gl::LazyAcc a0{}, a1{};   // sum of alpha^j f_j[i], reduced once (the powers are wave-uniform: scalar operands)
    for (unsigned j = j0; j < j1; ++j) {
        const gl::Ext a = load_ext(alpha_pows, j);
        const u64 v = polys[j][i];
        a0.mac((u32)v, (u32)(v >> 32), (u32)a.c0, (u32)(a.c0 >> 32));
        a1.mac((u32)v, (u32)(v >> 32), (u32)a.c1, (u32)(a.c1 >> 32));
    }
    part0[blockIdx.y * n + i] = gl::canon(a0.reduce());
    part1[blockIdx.y * n + i] = gl::canon(a1.reduce());
}
__global__ void __launch_bounds__(THREADS)
combine_finish_kernel(const u64* __restrict__ part0, const u64* __restrict__ part1, unsigned groups, size_t n,
                      u64* __restrict__ f0, u64* __restrict__ f1) {
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (i >= n) return;
    u64 a = 0, b = 0;
    for (unsigned g = 0; g < groups; ++g) {
        a = gl::add(a, part0[g * n + i]);
        b = gl::add(b, part1[g * n + i]);
    }
    f0[i] = a;
    f1[i] = b;
}

// Division by (X - z) as a suffix scan: q_k = z^-(k+1) * sum_{i>k} F_i z^i  (q_{n-1} = 0), then
// final_k = final_k * scale + q_k.  Two launches over 256-element chunks: chunk totals, then carry + local scan.
__global__ void __launch_bounds__(THREADS)
divide_totals_kernel(const u64* __restrict__ f0, const u64* __restrict__ f1, const u64* __restrict__ zpow, size_t n,
                     u64* __restrict__ totals) {
    __shared__ u64 sh[2 * THREADS];
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    gl::Ext t = gl::ext(0);
    if (i < n) t = gl::mul(gl::Ext{f0[i], f1[i]}, load_ext(zpow, i));
    const gl::Ext tot = block_sum(t, sh);
    if (threadIdx.x == 0) store_ext(totals, blockIdx.x, tot);
}
__global__ void __launch_bounds__(THREADS)
divide_finish_kernel(const u64* __restrict__ f0, const u64* __restrict__ f1, const u64* __restrict__ zpow,
                     const u64* __restrict__ zinvpow, const u64* __restrict__ totals, gl::Ext scale, size_t n,
                     u64* __restrict__ fin0, u64* __restrict__ fin1) {
    __shared__ u64 sh[2 * THREADS];
    __shared__ u64 carry_sh[2];
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    // carry = sum of the totals of all later chunks
    gl::Ext c = gl::ext(0);
    for (unsigned k = blockIdx.x + 1 + threadIdx.x; k < gridDim.x; k += THREADS) c = gl::add(c, load_ext(totals, k));
    const gl::Ext carry = block_sum(c, sh);
    if (threadIdx.x == 0) { carry_sh[0] = carry.c0; carry_sh[1] = carry.c1; }
    __syncthreads();
    // inclusive suffix scan of t_i inside the chunk
    gl::Ext t = gl::ext(0);
    if (i < n) t = gl::mul(gl::Ext{f0[i], f1[i]}, load_ext(zpow, i));
    sh[threadIdx.x] = t.c0;
    sh[THREADS + threadIdx.x] = t.c1;
    __syncthreads();
    for (unsigned d = 1; d < THREADS; d <<= 1) {
        gl::Ext v = gl::Ext{sh[threadIdx.x], sh[THREADS + threadIdx.x]};
        if (threadIdx.x + d < THREADS) v = gl::add(v, gl::Ext{sh[threadIdx.x + d], sh[THREADS + threadIdx.x + d]});
        __syncthreads();
        sh[threadIdx.x] = v.c0;
        sh[THREADS + threadIdx.x] = v.c1;
        __syncthreads();
    }
    if (i >= n) return;
    gl::Ext after = gl::Ext{carry_sh[0], carry_sh[1]};  // sum_{j > i} t_j
    if (threadIdx.x + 1 < THREADS) after = gl::add(after, gl::Ext{sh[threadIdx.x + 1], sh[THREADS + threadIdx.x + 1]});
    const gl::Ext q = i + 1 < n ? gl::mul(after, load_ext(zinvpow, i + 1)) : gl::ext(0);
    const gl::Ext f = gl::add(gl::mul(gl::Ext{fin0[i], fin1[i]}, scale), q);
    fin0[i] = f.c0;
    fin1[i] = f.c1;
}

__global__ void shift_up_kernel(const u64* in0, const u64* in1, u64* c0, u64* c1, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    c0[i] = i ? in0[i - 1] : 0;
    c1[i] = i ? in1[i - 1] : 0;
}

__global__ void __launch_bounds__(THREADS)
fold_kernel(const u64* __restrict__ in0, const u64* __restrict__ in1, size_t n_out, unsigned arity_bits, gl::Ext beta,
            u64* __restrict__ out0, u64* __restrict__ out1) {
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (i >= n_out) return;
    const unsigned arity = 1u << arity_bits;
    gl::Ext acc = gl::ext(0);
    for (unsigned j = arity; j-- > 0;) acc = gl::add(gl::mul(acc, beta), gl::Ext{in0[(i << arity_bits) + j], in1[(i << arity_bits) + j]});
    out0[i] = acc.c0;
    out1[i] = acc.c1;
}

// grid (n_trees, n_queries): one workgroup copies one (leaf, Merkle path) record
__global__ void __launch_bounds__(THREADS) open_queries_kernel(const OpenArgs* __restrict__ a, u64* __restrict__ out) {
    const OpenTree& t = a->trees[blockIdx.x];
    const size_t gidx = a->x_index[blockIdx.y] >> t.index_shift;
    u64* rec = out + blockIdx.y * a->record_words + t.out_off;
    if (gidx < t.leaf_lo || gidx >= t.leaf_hi) {  // owned by another rank
        for (unsigned w = threadIdx.x; w < t.leaf_len + 4 * t.n_siblings; w += THREADS) rec[w] = 0;
        return;
    }
    const size_t idx = gidx - t.leaf_lo;
    if (t.data1 == nullptr) {
        for (unsigned e = threadIdx.x; e < t.leaf_len; e += THREADS) rec[e] = t.data0[e * t.col_stride + idx];
    } else {
        for (unsigned e = threadIdx.x; e < t.leaf_len; e += THREADS)
            rec[e] = ((e & 1) ? t.data1 : t.data0)[(idx << t.arity_bits) + (e >> 1)];
    }
    for (unsigned w = threadIdx.x; w < 4 * t.n_siblings; w += THREADS) {
        const unsigned k = w >> 2;
        rec[t.leaf_len + w] = t.digests[t.level_off[k] + 4 * ((idx >> k) ^ 1) + (w & 3)];
    }
}
}  // namespace

void launch_ext_powers(hipStream_t s, const gl::Ext* points, unsigned count, size_t n, u64* out) {
    if (!n || !count) return;
    const unsigned per = (unsigned)std::min<size_t>(16, n & (~n + 1));   // 16 entries per thread (the largest power of two dividing n below that)
    const size_t lanes = n / per;
    PowerPoints pts{};
    for (unsigned t = 0; t < count && t < 4; ++t) {
        pts.z[t] = points[t];
        pts.stride[t] = gl::pow(points[t], lanes);
    }
    hipLaunchKernelGGL(ext_powers_kernel, dim3((unsigned)((lanes + 255) / 256), count), dim3(256), 0, s, pts, n, per, out);
}

// scratch-free two-step evaluation: partial sums live at the tail of `out` (caller provides [ncols*(1+chunks)][2])
void launch_eval_ext(hipStream_t s, const u64* coeffs, unsigned ncols, size_t n, size_t col_stride, const u64* zpow, u64* out) {
    const unsigned chunk_len = 4096;
    const unsigned chunks = (unsigned)((n + chunk_len - 1) / chunk_len);
    u64* partial = out + 2 * (size_t)ncols;
    hipLaunchKernelGGL(eval_partial_kernel, dim3(chunks, ncols), dim3(THREADS), 0, s, coeffs, n, col_stride, zpow, chunk_len, partial);
    hipLaunchKernelGGL(eval_finish_kernel, dim3((ncols + 63) / 64), dim3(64), 0, s, (const u64*)partial, chunks, ncols, out);
}

void launch_eval_ext_multi(hipStream_t s, const EvalSegments& segs, size_t n, u64* out) {
    const unsigned chunk_len = 4096;
    const unsigned chunks = (unsigned)((n + chunk_len - 1) / chunk_len);
    unsigned total = 0;
    for (unsigned k = 0; k < segs.count; ++k) total += segs.seg[k].ncols;
    u64* partial = out + 2 * (size_t)total;
    hipLaunchKernelGGL(eval_partial_multi_kernel, dim3(chunks, total), dim3(THREADS), 0, s, segs, n, chunk_len, partial);
    hipLaunchKernelGGL(eval_finish_kernel, dim3((total + 63) / 64), dim3(64), 0, s, (const u64*)partial, chunks, total, out);
}

void launch_combine(hipStream_t s, const u64* const* polys, unsigned n_polys, const u64* alpha_pows, size_t n, u64* f0, u64* f1,
                    u64* partial_scratch) {
    const unsigned groups = n_polys >= 64 ? COMBINE_GROUPS : 1;
    if (groups == 1) {
        hipLaunchKernelGGL(combine_kernel, dim3((n + THREADS - 1) / THREADS, 1), dim3(THREADS), 0, s, polys, n_polys, alpha_pows, n, f0, f1);
        return;
    }
    u64* p0 = partial_scratch;
    u64* p1 = partial_scratch + (size_t)groups * n;
    hipLaunchKernelGGL(combine_kernel, dim3((n + THREADS - 1) / THREADS, groups), dim3(THREADS), 0, s, polys, n_polys, alpha_pows, n, p0, p1);
    hipLaunchKernelGGL(combine_finish_kernel, dim3((n + THREADS - 1) / THREADS), dim3(THREADS), 0, s, (const u64*)p0, (const u64*)p1, groups, n,
                       f0, f1);
}

void launch_divide_accumulate(hipStream_t s, const u64* f0, const u64* f1, const u64* zpow, const u64* zinvpow, gl::Ext scale,
                              size_t n, u64* fin0, u64* fin1, u64* totals_scratch) {
    const unsigned chunks = (unsigned)((n + THREADS - 1) / THREADS);
    hipLaunchKernelGGL(divide_totals_kernel, dim3(chunks), dim3(THREADS), 0, s, f0, f1, zpow, n, totals_scratch);
    hipLaunchKernelGGL(divide_finish_kernel, dim3(chunks), dim3(THREADS), 0, s, f0, f1, zpow, zinvpow, (const u64*)totals_scratch, scale,
                       n, fin0, fin1);
}

void launch_shift_up(hipStream_t s, const u64* in0, const u64* in1, u64* out0, u64* out1, size_t n) {
    hipLaunchKernelGGL(shift_up_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in0, in1, out0, out1, n);
}

void launch_fold(hipStream_t s, const u64* in0, const u64* in1, size_t n_out, unsigned arity_bits, gl::Ext beta, u64* out0, u64* out1) {
    hipLaunchKernelGGL(fold_kernel, dim3((n_out + THREADS - 1) / THREADS), dim3(THREADS), 0, s, in0, in1, n_out, arity_bits, beta, out0, out1);
}

void launch_open_queries(hipStream_t s, const OpenArgs* d_args, unsigned n_trees, unsigned n_queries, u64* out) {
    hipLaunchKernelGGL(open_queries_kernel, dim3(n_trees, n_queries), dim3(THREADS), 0, s, d_args, out);
}
}  // namespace vpbs
