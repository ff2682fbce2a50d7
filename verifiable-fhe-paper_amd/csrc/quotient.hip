// Quotient polynomials, permutation-argument part, on gfx950.
// Replaces plonky2 0.2.0 plonk/prover.rs `compute_quotient_polys` + plonk/vanishing_poly.rs
// `eval_vanishing_poly_base_batch` (the L_0 (Z - 1) terms and `check_partial_products`), the Z_H division of
// plonk/plonk_common.rs `ZeroPolyOnCoset`, the coset iFFT and the split into degree-n chunks -- the stage between the
// Z/partial-products commitment and the quotient commitment of prove() (/root/reference/src/vtfhe/
// ivc_based_vpbs.rs:302,333,364).  The gate-constraint terms are evaluated by
// gates.hip; they enter here already alpha-folded per challenge (quotient_perm_kernel) or are joined in by quotient_combine_kernel
// when the gates ran concurrently with the permutation part.
//
// One thread per LDE point: it reads the committed LDE columns of the routed wires, the sigmas and the Z / partial
// products straight from the batches' HBM buffers (column-major, leaf order => coalesced), so nothing is downloaded
// (the reference's get_lde_values path moves 283 MB per step through the host).  Streaming, ~180 columns x 8 B per point.
#define GL_ASM_SCRATCH_LOW 1  // low asm scratch block: these kernels need few registers of their own (occupancy)
#include "kernels.h"
#include "poseidon.h"   // fold96: 7 x on residues

namespace vpbs {
namespace {
constexpr unsigned THREADS = 256;

struct QuotientConsts {
    u64 zh[8], zh_inv[8];  // Z_H on the 2^rate_bits cosets: 7^n w_8^r - 1, and its inverse
    u64 beta[4], gamma[4];
    u64 n_field;            // n as a field element
};

// L_0 on the coset in leaf order (ZeroPolyOnCoset::eval_l_0): Z_H(x) / (n (x - 1)).  Depends on the degree only, so a
// context computes it once (one field inversion per point) and the quotient kernel just reads it.
__global__ void __launch_bounds__(THREADS)
l0_table_kernel(const u64* __restrict__ roots_big, QuotientConsts k, unsigned log_n, unsigned rate_bits, u64* __restrict__ l0) {
    const unsigned log_big = log_n + rate_bits;
    const size_t big = (size_t)1 << log_big;
    const size_t j = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (j >= big) return;
    const unsigned t = gl::bitrev32((u32)j, log_big);
    const u64 wt = t < big / 2 ? roots_big[t] : gl::neg(roots_big[t - big / 2]);
    const u64 x = gl::mul(gl::GENERATOR, wt);
    l0[j] = gl::mul(k.zh[t & ((1u << rate_bits) - 1)], gl::inv(gl::mul(k.n_field, gl::sub(x, 1))));
}

// grid (8n / 256).  apow: [nc][n_terms + 1] powers of alpha_a.  q: [nc][8n] in leaf order.
// DEG: compile-time chunk size (8 = plonky2's quotient_degree_factor) so that the 16 loads of a chunk are issued
// together; DEG = 0 selects the generic run-time loop.  NCT: the number of challenges when it is plonky2's standard 2 (0 = run-time nc): the
// `c < nc` tests around every per-challenge operation then fold away, and with them the copies of the running products the compiler kept
// around each predicated update.
template <unsigned DEG, unsigned NCT>
__global__ void __launch_bounds__(THREADS)
quotient_perm_kernel(const u64* __restrict__ wires, const u64* __restrict__ sigmas, const u64* __restrict__ zs_pp,
                     const u64* __restrict__ roots_big, const u64* __restrict__ l0_table, const u64* __restrict__ gate_terms,
                     const u64* __restrict__ apow,
                     QuotientConsts k, unsigned n_routed, unsigned log_n, unsigned rate_bits, unsigned max_degree, unsigned nc_given,
                     size_t leaf_offset, size_t local_len, u64* __restrict__ q, int raw) {
    const unsigned nc = NCT ? NCT : nc_given;
    // The column arrays hold the leaves [leaf_offset, leaf_offset + local_len) only (a whole number of cosets; the full
    // LDE when unsharded): `big` below is their column stride and j the LOCAL leaf index.
    const unsigned log_big = log_n + rate_bits;
    const size_t big = local_len;
    const size_t j = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (j >= local_len) return;
    const size_t full = (size_t)1 << log_big;
    const unsigned t = gl::bitrev32((u32)(leaf_offset + j), log_big);                 // natural index of this leaf
    const unsigned t_next = (t + (1u << rate_bits)) & (unsigned)(full - 1);           // g * x: next trace row (same coset)
    const size_t j_next = gl::bitrev32(t_next, log_big) - leaf_offset;
    const u64 wt = t < full / 2 ? roots_big[t] : gl::neg(roots_big[t - full / 2]);    // w_{8n}^t
    const u64 x = gl::mul(gl::GENERATOR, wt);
    const unsigned r = t & ((1u << rate_bits) - 1);
    const u64 l0 = l0_table[leaf_offset + j];                                                       // ZeroPolyOnCoset::eval_l_0
    const unsigned n_chunks = (n_routed + max_degree - 1) / max_degree, num_prods = n_chunks - 1;
    const unsigned n_terms = nc + nc * n_chunks;
    u64 acc[4] = {0, 0, 0, 0};  // sum_i term_i alpha_a^i for each challenge a (nc <= 4): u64 RESIDUES, made canonical at the end
    // every per-challenge array is indexed with compile-time indices (loops unrolled to 4 and predicated): no scratch
    // (the arithmetic below keeps residues, not canonical values, wherever the next operation accepts them: fused multiply-adds
    // gl::mad_nc / gl::dot2_nc with one reduction, products without the final conditional subtraction -- 82 instead of 109
    // instructions per (routed wire, challenge); the stored values are canonical and unchanged)
    auto add_term = [&](unsigned i, u64 term) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if ((unsigned)a < nc) acc[a] = gl::mad_nc(term, apow[a * (n_terms + 1) + i], acc[a]);
    };
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if ((unsigned)c < nc) add_term(c, gl::mul(l0, gl::sub(zs_pp[(size_t)c * big + j], 1)));
    // numerators / denominators: every column value is loaded once and used for all challenges
    u64 sid[4];  // beta_c * k_j * x, k_j = 7^j
#pragma unroll
    for (int c = 0; c < 4; ++c) sid[c] = (unsigned)c < nc ? gl::mul(k.beta[c], x) : 0;
    for (unsigned kk = 0; kk < n_chunks; ++kk) {
        u64 num[4] = {1, 1, 1, 1}, den[4] = {1, 1, 1, 1};
        auto absorb = [&](u64 w, u64 s) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if ((unsigned)c >= nc) continue;
                const u64 wg = gl::add_a(w, k.gamma[c]);                     // a residue: shared by numerator and denominator
                num[c] = gl::mul_nc(num[c], gl::add_a(wg, sid[c]));          // w + beta k_j x + gamma
                den[c] = gl::mul_nc(den[c], gl::mad_nc(k.beta[c], s, wg));   // w + beta sigma_j + gamma, one reduction
                sid[c] = poseidon::fold96((u64)(u32)sid[c] * 7u, (u64)(u32)(sid[c] >> 32) * 7u);   // 7 x on a residue: two multiply-adds and a fold
            }
        };
        if (DEG != 0 && (kk + 1) * DEG <= n_routed) {
            u64 wv[DEG ? DEG : 1], sv[DEG ? DEG : 1];
#pragma unroll
            for (unsigned u = 0; u < DEG; ++u) {
                wv[u] = wires[(size_t)(kk * DEG + u) * big + j];
                sv[u] = sigmas[(size_t)(kk * DEG + u) * big + j];
            }
#pragma unroll
            for (unsigned u = 0; u < DEG; ++u) absorb(wv[u], sv[u]);
        } else {
            for (unsigned col = kk * max_degree; col < (kk + 1) * max_degree && col < n_routed; ++col)
                absorb(wires[(size_t)col * big + j], sigmas[(size_t)col * big + j]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if ((unsigned)c >= nc) continue;
            const u64* zc = zs_pp + (size_t)c * big;
            const u64* ppc = zs_pp + ((size_t)nc + (size_t)c * num_prods) * big;
            const u64 prev = kk == 0 ? zc[j] : ppc[(size_t)(kk - 1) * big + j];
            const u64 next = kk == num_prods ? zc[j_next] : ppc[(size_t)kk * big + j];
            add_term(nc + c * n_chunks + kk, gl::dot2_nc(prev, num[c], gl::neg(next), den[c]));  // check_partial_products: prev num - next den
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if ((unsigned)a >= nc) continue;
        u64 v = gl::canon(acc[a]);
        if (raw) {  // the gate terms and the division by Z_H are applied by quotient_combine_kernel (the gates run concurrently)
            q[(size_t)a * big + j] = v;
            continue;
        }
        if (gate_terms) v = gl::add(v, gl::mul(gate_terms[(size_t)a * big + j], apow[a * (n_terms + 1) + n_terms]));
        q[(size_t)a * big + j] = gl::mul(v, k.zh_inv[r]);
    }
}

// q[a][j] <- (q[a][j] + alpha_a^(n_terms) * (g0 + g1 + g2)[a][j]) / Z_H(x_j): joins the permutation part (raw) with the gate terms of
// up to three lanes (null = unused)
struct CombineConsts {
    u64 zh_inv[8], apow_last[4];
};
__global__ void __launch_bounds__(THREADS)
quotient_combine_kernel(u64* __restrict__ q, const u64* __restrict__ g0, const u64* __restrict__ g1, const u64* __restrict__ g2,
                        CombineConsts k, unsigned log_big, unsigned rate_bits, size_t leaf_offset, size_t local_len) {
    const size_t j = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (j >= local_len) return;
    const unsigned a = blockIdx.y;
    const unsigned t = gl::bitrev32((u32)(leaf_offset + j), log_big);
    const size_t i = (size_t)a * local_len + j;
    u64 g = g0 ? g0[i] : 0;
    if (g1) g = gl::add(g, g1[i]);
    if (g2) g = gl::add(g, g2[i]);
    q[i] = gl::mul(gl::add(q[i], gl::mul(g, k.apow_last[a])), k.zh_inv[t & ((1u << rate_bits) - 1)]);
}

// the same with the gate terms as n_planes planes [plane][nc][local_len] (the one-launch gate kernel's items)
__global__ void __launch_bounds__(THREADS)
quotient_combine_planes_kernel(u64* __restrict__ q, const u64* __restrict__ planes, unsigned n_planes, CombineConsts k, unsigned log_big,
                               unsigned rate_bits, unsigned nc, size_t leaf_offset, size_t local_len) {
    const size_t j = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (j >= local_len) return;
    const unsigned a = blockIdx.y;
    const unsigned t = gl::bitrev32((u32)(leaf_offset + j), log_big);
    const size_t i = (size_t)a * local_len + j, plane = (size_t)nc * local_len;
    u64 g = planes[i];
    for (unsigned p = 1; p < n_planes; ++p) g = gl::add(g, planes[p * plane + i]);
    q[i] = gl::mul(gl::add(q[i], gl::mul(g, k.apow_last[a])), k.zh_inv[t & ((1u << rate_bits) - 1)]);
}

// out[a][t] = value of challenge a at leaf bitrev(t)  (leaf order -> natural order).  `in` is rank-major
// [world][nc][local_len] (the layout an all-gather of per-rank [nc][local_len] buffers produces; world = 1: [nc][len]).
__global__ void bitrev_copy_kernel(const u64* __restrict__ in, u64* __restrict__ out, unsigned log_len, size_t local_len, unsigned nc) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t len = (size_t)1 << log_len;
    if (t >= len) return;
    const size_t j = gl::bitrev32((u32)t, log_len);
    const size_t rank = j / local_len, jl = j - rank * local_len;
    out[blockIdx.y * len + t] = in[(rank * nc + blockIdx.y) * local_len + jl];
}
__global__ void mul_table_kernel(u64* __restrict__ data, const u64* __restrict__ table, size_t len) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < len) data[blockIdx.y * len + i] = gl::mul(data[blockIdx.y * len + i], table[i]);
}
}  // namespace

static QuotientConsts make_consts(unsigned log_n, unsigned rate_bits) {
    QuotientConsts k{};
    const size_t n = (size_t)1 << log_n;
    const u64 seven_n = gl::pow(gl::GENERATOR, n), w8 = gl::root_of_unity(rate_bits);
    for (unsigned r = 0; r < (1u << rate_bits) && r < 8; ++r) {
        k.zh[r] = gl::sub(gl::mul(seven_n, gl::pow(w8, r)), 1);
        k.zh_inv[r] = gl::inv(k.zh[r]);
    }
    k.n_field = (u64)n;
    return k;
}

void launch_l0_table(hipStream_t s, const u64* roots_big, unsigned log_n, unsigned rate_bits, u64* l0) {
    const size_t big = (size_t)1 << (log_n + rate_bits);
    hipLaunchKernelGGL(l0_table_kernel, dim3((unsigned)((big + THREADS - 1) / THREADS)), dim3(THREADS), 0, s, roots_big,
                       make_consts(log_n, rate_bits), log_n, rate_bits, l0);
}

void launch_quotient_values(hipStream_t s, const u64* wires_lde, const u64* sigmas_lde, const u64* zs_pp_lde, const u64* roots_big,
                            const u64* l0_table, const u64* d_gate_terms, const u64* d_apow, const u64* betas, const u64* gammas,
                            unsigned n_routed, unsigned log_n, unsigned rate_bits, unsigned max_degree, unsigned nc, size_t leaf_offset,
                            size_t local_len, u64* q_leaf_local, bool raw) {
    QuotientConsts k = make_consts(log_n, rate_bits);
    for (unsigned c = 0; c < nc; ++c) {
        k.beta[c] = betas[c];
        k.gamma[c] = gammas[c];
    }
    const dim3 grid((unsigned)((local_len + THREADS - 1) / THREADS));
    const auto kernel = max_degree == 8 ? (nc == 2 ? quotient_perm_kernel<8, 2> : quotient_perm_kernel<8, 0>) : quotient_perm_kernel<0, 0>;
    hipLaunchKernelGGL(kernel, grid, dim3(THREADS), 0, s, wires_lde, sigmas_lde, zs_pp_lde, roots_big, l0_table, d_gate_terms, d_apow, k, n_routed,
                       log_n, rate_bits, max_degree, nc, leaf_offset, local_len, q_leaf_local, raw ? 1 : 0);
}

void launch_quotient_combine(hipStream_t s, u64* q_local, const u64* g0, const u64* g1, const u64* g2, const u64* apow_last, unsigned log_n,
                             unsigned rate_bits, unsigned nc, size_t leaf_offset, size_t local_len) {
    const QuotientConsts qc = make_consts(log_n, rate_bits);
    CombineConsts k{};
    for (unsigned r = 0; r < 8; ++r) k.zh_inv[r] = qc.zh_inv[r];
    for (unsigned a = 0; a < nc; ++a) k.apow_last[a] = apow_last[a];
    hipLaunchKernelGGL(quotient_combine_kernel, dim3((unsigned)((local_len + THREADS - 1) / THREADS), nc), dim3(THREADS), 0, s, q_local, g0, g1, g2, k,
                       log_n + rate_bits, rate_bits, leaf_offset, local_len);
}

void launch_quotient_combine_planes(hipStream_t s, u64* q_local, const u64* planes, unsigned n_planes, const u64* apow_last, unsigned log_n,
                                    unsigned rate_bits, unsigned nc, size_t leaf_offset, size_t local_len) {
    const QuotientConsts qc = make_consts(log_n, rate_bits);
    CombineConsts k{};
    for (unsigned r = 0; r < 8; ++r) k.zh_inv[r] = qc.zh_inv[r];
    for (unsigned a = 0; a < nc; ++a) k.apow_last[a] = apow_last[a];
    hipLaunchKernelGGL(quotient_combine_planes_kernel, dim3((unsigned)((local_len + THREADS - 1) / THREADS), nc), dim3(THREADS), 0, s, q_local, planes,
                       n_planes, k, log_n + rate_bits, rate_bits, nc, leaf_offset, local_len);
}

void launch_quotient_finish(hipStream_t s, const u64* q_gathered, size_t local_len, const u64* inv_roots_big, const u64* unshift_table,
                            unsigned log_n, unsigned rate_bits, unsigned nc, u64* q_nat, u64* scratch, u64* out_coeffs) {
    const unsigned log_big = log_n + rate_bits;
    const size_t big = (size_t)1 << log_big;
    // PolynomialValues::coset_ifft(7): natural order -> iNTT of size 8n -> coefficient i times 7^-i; chunk m of challenge a is
    // out_coeffs[(a * 8 + m) * n ..]
    hipLaunchKernelGGL(bitrev_copy_kernel, dim3((unsigned)((big + 255) / 256), nc), dim3(256), 0, s, q_gathered, q_nat, log_big, local_len, nc);
    launch_intt(s, q_nat, out_coeffs, scratch, inv_roots_big, nc, log_big);
    hipLaunchKernelGGL(mul_table_kernel, dim3((unsigned)((big + 255) / 256), nc), dim3(256), 0, s, out_coeffs, unshift_table, big);
}
}  // namespace vpbs
