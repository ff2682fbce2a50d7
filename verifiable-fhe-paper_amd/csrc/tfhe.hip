// Native TFHE data path of one vPBS step on gfx950, batched over independent accumulators: rotation by the mod-switched
// mask, CMUX difference, signed base-2^LOGB decomposition, forward negacyclic NTT of the top ELL limbs, multiply-accumulate
// with the (NTT-domain) GGSW rows, inverse NTT, CMUX add.  This is what the step circuit of the reference computes
// in-circuit (/root/reference/src/vtfhe/ivc_based_vpbs.rs:99-125, mod.rs:80-136, glwe_poly.rs:28-50,132-166,
// glev_ct.rs:92-110, ggsw_ct.rs:98-112) and therefore the accumulator every step proof exposes as public inputs: the
// native core of witness generation.  The negacyclic transform is the reference's
// (crypto/poly.rs:9-64, tables per src/ntt/params_{N}.rs; pinned by TESTG/TESTGHAT).
// One workgroup per (instance, polynomial); everything for a polynomial stays in LDS (ELL x N x 8 B <= 128 KiB).
#define GL_ASM_SCRATCH_LOW 1  // low asm scratch block: these kernels need few registers of their own (occupancy)
#include "kernels.h"

namespace vpbs {
namespace {
constexpr unsigned THREADS = 256;

// rotation amount in [0, 2N]: top log2(2N) bits of the canonical mask, rounded with the next bit (mod.rs:85-106)
__device__ __forceinline__ unsigned mod_switch(u64 mask, unsigned log_n_ring) {
    const unsigned log2n = log_n_ring + 1;
    return (unsigned)(mask >> (64 - log2n)) + (unsigned)((mask >> (64 - log2n - 1)) & 1);
}
// coefficient i of poly * X^shift mod X^N + 1, 0 <= shift <= 2N
__device__ __forceinline__ u64 rotated_coeff(const u64* __restrict__ poly, unsigned n, unsigned shift, unsigned i) {
    const unsigned src = (i + 2 * n - shift) & (2 * n - 1);  // exponent whose image is i, in [0, 2N)
    const u64 c = poly[src & (n - 1)];
    return src >= n ? gl::neg(c) : c;
}

// in-place forward negacyclic NTT of `cnt` polynomials stored back to back in LDS (crypto/poly.rs:9-34)
__device__ void negacyclic_fw_lds(u64* t, unsigned n, unsigned cnt, const u64* __restrict__ roots) {
    for (unsigned m = 1; m < n; m <<= 1) {
        const unsigned len = n / (2 * m);
        for (unsigned k = threadIdx.x; k < cnt * (n / 2); k += THREADS) {
            const unsigned poly = k / (n / 2), kk = k % (n / 2);
            const unsigned i = kk / len, j = 2 * i * len + (kk % len);
            u64* p = t + poly * n;
            const u64 u = p[j], v = gl::mul(p[j + len], roots[m + i]);
            p[j] = gl::add(u, v);
            p[j + len] = gl::sub(u, v);
        }
        __syncthreads();
    }
}
// in-place inverse (crypto/poly.rs:36-64), including the multiplication by N^-1
__device__ void negacyclic_bw_lds(u64* t, unsigned n, const u64* __restrict__ invroots, u64 ninv) {
    for (unsigned m = n >> 1; m >= 1; m >>= 1) {
        const unsigned len = n / (2 * m);
        for (unsigned k = threadIdx.x; k < n / 2; k += THREADS) {
            const unsigned i = k / len, j = 2 * i * len + (k % len);
            const u64 u = t[j], v = t[j + len];
            t[j] = gl::add(u, v);
            t[j + len] = gl::mul(gl::sub(u, v), invroots[m + i]);
        }
        __syncthreads();
    }
    for (unsigned i = threadIdx.x; i < n; i += THREADS) t[i] = gl::mul(t[i], ninv);
    __syncthreads();
}

struct TfheShape {
    unsigned log_n, K, ELL, LOGB;
};

// grid (batch, K): decompose polynomial p of the external-product input and NTT its top ELL limbs.
// limbs_hat: [batch][K][ELL][N]
__global__ void __launch_bounds__(THREADS)
br_decompose_ntt_kernel(const u64* __restrict__ acc_in, const u64* __restrict__ masks, const u64* __restrict__ roots, TfheShape sh,
                        int last_step, u64* __restrict__ limbs_hat) {
    extern __shared__ __align__(16) u64 lds[];  // [ELL][N]
    const unsigned n = 1u << sh.log_n, b = blockIdx.x, p = blockIdx.y;
    const u64* poly = acc_in + ((size_t)b * sh.K + p) * n;
    const unsigned shift = mod_switch(masks[b], sh.log_n);
    const unsigned nl = (64 + sh.LOGB - 1) / sh.LOGB, tb = nl * sh.LOGB;
    for (unsigned i = threadIdx.x; i < n; i += THREADS) {
        // xprod_in = last_step ? acc : rotate(acc, mask) - acc     (ivc_based_vpbs.rs:113-116)
        const u64 x = last_step ? poly[i] : gl::sub(rotated_coeff(poly, n, shift, i), poly[i]);
        // decompose (glwe_poly.rs:28-50)
        const unsigned sgn = tb <= 64 ? (unsigned)((x >> (tb - 1)) & 1) : 0;
        const u64 xc = sgn ? gl::neg(x) : x;
        unsigned carry = 0;
        for (unsigned l = 0; l < nl; ++l) {
            const unsigned lo_bit = l * sh.LOGB;
            const u64 k = (lo_bit < 64 ? (xc >> lo_bit) : 0) & (((u64)1 << sh.LOGB) - 1);
            const u64 kw = k + carry;
            carry = (unsigned)((k >> (sh.LOGB - 1)) & 1);
            const u64 bal = gl::sub(kw, (u64)carry << sh.LOGB);  // k_w_carry - carry * B in the field
            if (l + sh.ELL >= nl) lds[(l + sh.ELL - nl) * n + i] = sgn ? gl::neg(bal) : bal;
        }
    }
    __syncthreads();
    negacyclic_fw_lds(lds, n, sh.ELL, roots);
    u64* dst = limbs_hat + (((size_t)b * sh.K + p) * sh.ELL) * n;
    for (unsigned i = threadIdx.x; i < sh.ELL * n; i += THREADS) dst[i] = lds[i];
}

// grid (batch, K): output polynomial r.  ggsw: [K][ELL][K][N] (shared) or [batch][K][ELL][K][N]
__global__ void __launch_bounds__(THREADS)
br_mac_intt_kernel(const u64* __restrict__ acc_in, const u64* __restrict__ masks, const u64* __restrict__ limbs_hat,
                   const u64* __restrict__ ggsw, size_t ggsw_instance_stride, const u64* __restrict__ invroots, u64 ninv, TfheShape sh,
                   int first_step, int last_step, u64* __restrict__ acc_out) {
    extern __shared__ __align__(16) u64 lds[];  // [N]
    const unsigned n = 1u << sh.log_n, b = blockIdx.x, r = blockIdx.y;
    const u64* poly = acc_in + ((size_t)b * sh.K + r) * n;
    u64* out = acc_out + ((size_t)b * sh.K + r) * n;
    if (first_step) {  // body step: rotate by -mask only (ivc_based_vpbs.rs:106-111,122)
        const unsigned shift = mod_switch(gl::neg(masks[b]), sh.log_n);
        for (unsigned i = threadIdx.x; i < n; i += THREADS) out[i] = rotated_coeff(poly, n, shift, i);
        return;
    }
    const u64* g = ggsw + b * ggsw_instance_stride;
    for (unsigned i = threadIdx.x; i < n; i += THREADS) {
        // glev_muls[K-1] - sum_{p < K-1} glev_muls[p]   (ggsw_ct.rs:109-111), each a vec_inner over the ELL limbs
        u64 pos = 0, negs = 0;
        for (unsigned p = 0; p < sh.K; ++p) {
            u64 acc = 0;
            for (unsigned l = 0; l < sh.ELL; ++l)
                acc = gl::add(acc, gl::mul(limbs_hat[(((size_t)b * sh.K + p) * sh.ELL + l) * n + i],
                                           g[(((size_t)p * sh.ELL + l) * sh.K + r) * n + i]));
            if (p + 1 == sh.K) pos = acc; else negs = gl::add(negs, acc);
        }
        lds[i] = gl::sub(pos, negs);
    }
    __syncthreads();
    negacyclic_bw_lds(lds, n, invroots, ninv);
    for (unsigned i = threadIdx.x; i < n; i += THREADS) out[i] = last_step ? lds[i] : gl::add(lds[i], poly[i]);  // CMUX add
}
}  // namespace

void launch_blind_rotate_step(hipStream_t s, const u64* acc_in, const u64* masks, const u64* ggsw, size_t ggsw_instance_stride,
                              const u64* roots, const u64* invroots, u64 ninv, unsigned log_n, unsigned K, unsigned ELL, unsigned LOGB,
                              unsigned batch, int first_step, int last_step, u64* limbs_hat, u64* acc_out) {
    const TfheShape sh{log_n, K, ELL, LOGB};
    const size_t n = (size_t)1 << log_n;
    if (!first_step)
        hipLaunchKernelGGL(br_decompose_ntt_kernel, dim3(batch, K), dim3(THREADS), ELL * n * sizeof(u64), s, acc_in, masks, roots, sh,
                           last_step, limbs_hat);
    hipLaunchKernelGGL(br_mac_intt_kernel, dim3(batch, K), dim3(THREADS), n * sizeof(u64), s, acc_in, masks, (const u64*)limbs_hat, ggsw,
                       ggsw_instance_stride, invroots, ninv, sh, first_step, last_step, acc_out);
}
}  // namespace vpbs
