// Permutation-argument Z polynomials and partial products on gfx950.
// Replaces plonky2 0.2.0 plonk/prover.rs `all_wires_permutation_partial_products` /
// `wires_permutation_partial_products_and_zs` (+ `quotient_chunk_products`, `partial_products_and_z_gx`), the stage
// between the wires commitment and the Z/partial-products commitment of prove() --
// /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364; SURVEY.md 8a row a12, 8f-1, Appendix A.9.
//
// Per row i and challenge c: num_j = w_j + beta k_j x_i + gamma, den_j = w_j + beta sigma_j + gamma (j < n_routed,
// k_j = 7^j, x_i = w_n^i); chunk quotient c_k = prod(num) / prod(den) over 8 consecutive j (one field inversion per
// chunk instead of a per-element batch inverse: same field element); Z(w^0) = 1, Z(w^(i+1)) = Z(w^i) prod_k c_k;
// pp_k(i) = Z(w^i) c_0 .. c_k.  The running product over rows is a two-level multiplicative scan.  Streaming over the
// wire and sigma columns (coalesced across rows); exact arithmetic => bit-identical to the sequential reference.
#define GL_ASM_SCRATCH_LOW 1  // low asm scratch block: these kernels need few registers of their own (occupancy)
#include "kernels.h"
#include "poseidon.h"   // fold96: 7 x on residues

namespace vpbs {
namespace {
constexpr unsigned THREADS = 256;

// grid (n / 256, num_challenges, n_chunks): one thread = one chunk quotient prod(num) / prod(den) of one row (one field
// inversion each; 10x the threads of a row-per-thread kernel, which at n = 2^15 would run at one wave per SIMD).
__global__ void __launch_bounds__(THREADS)
pp_chunk_kernel(const u64* __restrict__ wires, const u64* __restrict__ sigmas, const u64* __restrict__ roots, unsigned n_routed,
                unsigned log_n, unsigned max_degree, const u64* __restrict__ betas, const u64* __restrict__ gammas,
                u64* __restrict__ chunk_q, unsigned* __restrict__ zero_flag) {
    const size_t n = (size_t)1 << log_n;
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    const unsigned c = blockIdx.y, k = blockIdx.z, n_chunks = gridDim.z;
    if (i >= n) return;
    const u64 beta = betas[c], gamma = gammas[c];
    const u64 x = i < n / 2 ? roots[i] : gl::neg(roots[i - n / 2]);  // w^i (w^(n/2) = -1)
    u64 t = gl::mul(gl::mul(beta, x), gl::pow(gl::GENERATOR, (u64)k * max_degree));  // beta * k_j * x at j = k * max_degree
    u64 num = 1, den = 1;
    for (unsigned j = k * max_degree; j < (k + 1) * max_degree && j < n_routed; ++j) {
        const u64 w = wires[(size_t)j * n + i];
        num = gl::mul(num, gl::add(gl::add(w, t), gamma));
        den = gl::mul(den, gl::add(gl::add(w, gl::mul(beta, sigmas[(size_t)j * n + i])), gamma));
        t = gl::mul7(t);
    }
    if (den == 0) atomicOr(zero_flag, 1u);  // plonky2's batch inverse would panic here
    chunk_q[((size_t)c * n_chunks + k) * n + i] = gl::mul(num, gl::inv(den));
}

// grid (n / 256, num_challenges): cumulative products of the row's chunk quotients; the first num_prods go to `pp` (scaled by
// Z later), the full row product to `rowprod`.
__global__ void __launch_bounds__(THREADS)
pp_row_kernel(const u64* __restrict__ chunk_q, size_t n, unsigned n_chunks, u64* __restrict__ pp, u64* __restrict__ rowprod) {
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    const unsigned c = blockIdx.y;
    if (i >= n) return;
    const unsigned num_prods = n_chunks - 1;
    u64 run = 1;
    for (unsigned k = 0; k < n_chunks; ++k) {
        run = gl::mul(run, chunk_q[((size_t)c * n_chunks + k) * n + i]);
        if (k < num_prods) pp[((size_t)c * num_prods + k) * n + i] = run;
    }
    rowprod[(size_t)c * n + i] = run;
}

// The standard shape (80 routed wires in 10 chunks of 8) in ONE kernel, one thread per (row, challenge): the ten chunk quotients share ONE
// field inversion (Montgomery's trick: prefix products of the denominators, one inverse, back-substitution -- what plonky2's
// batch_multiplicative_inverse does across rows), the running products of pp_row_kernel follow in registers, and everything in between is
// computed on u64 residues (gl::add_a / mul_nc / mad_nc; stored values canonical).  An inversion is 72 multiplications against the chunk's own
// 24: 2.0 k -> 0.78 k instructions per (row, challenge, chunk), and chunk_q is neither written nor read.  Field results are exact, so the
// values equal those of the three-kernel path bit for bit (a zero denominator raises the flag, and the step fails, on either path).
template <unsigned DEG, unsigned CHUNKS>
__global__ void __launch_bounds__(THREADS)
pp_rows_kernel(const u64* __restrict__ wires, const u64* __restrict__ sigmas, const u64* __restrict__ roots, unsigned log_n,
               const u64* __restrict__ betas, const u64* __restrict__ gammas, u64* __restrict__ pp, u64* __restrict__ rowprod,
               unsigned* __restrict__ zero_flag) {
    const size_t n = (size_t)1 << log_n;
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    const unsigned c = blockIdx.y;
    if (i >= n) return;
    const u64 beta = betas[c], gamma = gammas[c];
    const u64 x = i < n / 2 ? roots[i] : gl::neg(roots[i - n / 2]);
    u64 t = gl::mul_nc(beta, x);   // beta k_j x, k_j = 7^j: a residue
    u64 num[CHUNKS], den[CHUNKS];
#pragma unroll
    for (unsigned k = 0; k < CHUNKS; ++k) {
        u64 wv[DEG], sv[DEG];
#pragma unroll
        for (unsigned u = 0; u < DEG; ++u) {
            wv[u] = wires[(size_t)(k * DEG + u) * n + i];
            sv[u] = sigmas[(size_t)(k * DEG + u) * n + i];
        }
        u64 nu = 0, de = 0;
#pragma unroll
        for (unsigned u = 0; u < DEG; ++u) {
            const u64 wg = gl::add_a(wv[u], gamma);
            const u64 a = gl::add_a(wg, t), b = gl::mad_nc(beta, sv[u], wg);
            nu = u ? gl::mul_nc(nu, a) : a;
            de = u ? gl::mul_nc(de, b) : b;
            t = poseidon::fold96((u64)(u32)t * 7u, (u64)(u32)(t >> 32) * 7u);
        }
        num[k] = nu;
        den[k] = de;
    }
    u64 pre[CHUNKS];
    pre[0] = den[0];
#pragma unroll
    for (unsigned k = 1; k < CHUNKS; ++k) pre[k] = gl::mul_nc(pre[k - 1], den[k]);
    if (gl::canon(pre[CHUNKS - 1]) == 0) atomicOr(zero_flag, 1u);  // plonky2's batch inverse would panic here
    u64 inv_rest = gl::inv(pre[CHUNKS - 1]);   // 1 / (den[0] .. den[k]) as k runs down
#pragma unroll
    for (unsigned k = CHUNKS - 1; k > 0; --k) {
        num[k] = gl::mul_nc(num[k], gl::mul_nc(inv_rest, pre[k - 1]));   // num[k] / den[k]
        inv_rest = gl::mul_nc(inv_rest, den[k]);
    }
    u64 run = gl::mul_nc(num[0], inv_rest);
    u64* out = pp + (size_t)c * (CHUNKS - 1) * n + i;
#pragma unroll
    for (unsigned k = 0; k + 1 < CHUNKS; ++k) {
        if (k) run = gl::mul_nc(run, num[k]);
        out[(size_t)k * n] = gl::canon(run);
    }
    rowprod[(size_t)c * n + i] = gl::mul(run, num[CHUNKS - 1]);
}

__device__ u64 block_product(u64 v, u64* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (unsigned d = THREADS / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] = gl::mul(sh[threadIdx.x], sh[threadIdx.x + d]);
        __syncthreads();
    }
    const u64 r = sh[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(THREADS)
pp_block_prod_kernel(const u64* __restrict__ rowprod, size_t n, u64* __restrict__ block_prod) {
    __shared__ u64 sh[THREADS];
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    const u64 v = i < n ? rowprod[blockIdx.y * n + i] : 1;
    const u64 p = block_product(v, sh);
    if (threadIdx.x == 0) block_prod[blockIdx.y * gridDim.x + blockIdx.x] = p;
}

// Z(w^i) = product of the row products of all earlier rows; then scale the row's partial products by Z(w^i)
__global__ void __launch_bounds__(THREADS)
pp_finish_kernel(const u64* __restrict__ rowprod, const u64* __restrict__ block_prod, size_t n, unsigned num_prods,
                 unsigned num_challenges, u64* __restrict__ out) {
    __shared__ u64 sh[THREADS];
    const unsigned c = blockIdx.y;
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    // carry = product of the earlier blocks
    u64 acc = 1;
    for (unsigned b = threadIdx.x; b < blockIdx.x; b += THREADS) acc = gl::mul(acc, block_prod[c * gridDim.x + b]);
    const u64 carry = block_product(acc, sh);
    // inclusive prefix product inside the block (Hillis-Steele)
    sh[threadIdx.x] = i < n ? rowprod[c * n + i] : 1;
    __syncthreads();
    for (unsigned d = 1; d < THREADS; d <<= 1) {
        u64 v = sh[threadIdx.x];
        if (threadIdx.x >= d) v = gl::mul(v, sh[threadIdx.x - d]);
        __syncthreads();
        sh[threadIdx.x] = v;
        __syncthreads();
    }
    if (i >= n) return;
    const u64 z = threadIdx.x == 0 ? carry : gl::mul(carry, sh[threadIdx.x - 1]);  // exclusive: rows before i
    out[(size_t)c * n + i] = z;                                                      // batch order: Z polynomials first
    u64* pp = out + ((size_t)num_challenges + (size_t)c * num_prods) * n;
    for (unsigned k = 0; k < num_prods; ++k) pp[(size_t)k * n + i] = gl::mul(pp[(size_t)k * n + i], z);
}
}  // namespace

void launch_partial_products(hipStream_t s, const u64* wires, const u64* sigmas, const u64* roots, unsigned n_routed, unsigned log_n,
                             unsigned max_degree, const u64* d_betas, const u64* d_gammas, unsigned num_challenges, u64* out,
                             u64* scratch, unsigned* d_zero_flag) {
    const size_t n = (size_t)1 << log_n;
    const unsigned n_chunks = (n_routed + max_degree - 1) / max_degree, num_prods = n_chunks - 1;
    const unsigned blocks = (unsigned)((n + THREADS - 1) / THREADS);
    u64* rowprod = scratch;                                         // [nc][n]
    u64* block_prod = scratch + (size_t)num_challenges * n;          // [nc][blocks]
    u64* chunk_q = block_prod + (size_t)num_challenges * blocks;     // [nc][n_chunks][n]
    u64* pp = out + (size_t)num_challenges * n;
    if (max_degree == 8 && n_routed == 80) {
        hipLaunchKernelGGL((pp_rows_kernel<8, 10>), dim3(blocks, num_challenges), dim3(THREADS), 0, s, wires, sigmas, roots, log_n, d_betas, d_gammas, pp,
                           rowprod, d_zero_flag);
    } else {
    hipLaunchKernelGGL(pp_chunk_kernel, dim3(blocks, num_challenges, n_chunks), dim3(THREADS), 0, s, wires, sigmas, roots, n_routed, log_n,
                       max_degree, d_betas, d_gammas, chunk_q, d_zero_flag);
    hipLaunchKernelGGL(pp_row_kernel, dim3(blocks, num_challenges), dim3(THREADS), 0, s, (const u64*)chunk_q, n, n_chunks, pp, rowprod);
    }
    hipLaunchKernelGGL(pp_block_prod_kernel, dim3(blocks, num_challenges), dim3(THREADS), 0, s, (const u64*)rowprod, n, block_prod);
    hipLaunchKernelGGL(pp_finish_kernel, dim3(blocks, num_challenges), dim3(THREADS), 0, s, (const u64*)rowprod, (const u64*)block_prod, n,
                       num_prods, num_challenges, out);
}
}  // namespace vpbs
