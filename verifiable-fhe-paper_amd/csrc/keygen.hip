// Seeded TFHE key / ciphertext generation on gfx950: what /root/reference/src/main.rs:40-52 draws from unseeded RNGs --
//   Glwe::partial_key(n) / flatten_partial_key (crypto/glwe.rs:19-40), Glwe::key_gen (:15-17), compute_bsk (crypto/mod.rs:29-45 =
//   Ggsw::encrypt(s_glwe, constant(s_i)).ntt_forward(), crypto/ggsw.rs:26-36, crypto/glev.rs:26-38, crypto/glwe.rs:49-57),
//   Ggsw::compute_ksk (crypto/ggsw.rs:38-48), lwe::encrypt (crypto/lwe.rs:55-64), get_testv / get_delta (crypto/mod.rs:17-27, lwe.rs:50-52)
// -- from ONE 64-bit seed, so that the prover, the CPU oracle and a patched reference can be fed identical keys and inputs
// (north_star: "identical synthetic GLWE keys/inputs"; the reference has no hook for that).
//
// Generator (restated bit for bit by tests/tfhe_oracle.py `Seeded`):
//   mix64        = the splitmix64 finaliser
//   draw(tag, i) = mix64(mix64(seed + G (tag + 1)) + G (i + 1)),  G = 0x9E3779B97F4A7C15           (counter based: any element on its own)
//   field        = draw >= p ? draw - p : draw                      (bias 2^-32, as `F::rand` has none: keys for benchmarking, not secrets)
//   bit          = draw & 1                                         (crypto/poly.rs:80 `random::<u64>() % 2`)
//   noise        = floor(T M / 2^32) mod p,  T = sum of the twelve 32-bit halves of draw(tag, 6 i .. 6 i + 5) - 6 2^32,  M = floor(sigma q + 1/2)
//                  -- an Irwin-Hall(12) stand-in for `Normal(0, sigma q).sample().round()` (crypto/lwe.rs:43-47): mean 0, standard deviation M,
//                  support +-6 sigma, integer arithmetic only, hence identical on every host and device (a libm Box-Muller would not be).
// The GLWE encryptions are generated directly in the NTT domain the circuit consumes:  ntt(a), ntt(a s + e + m) = ntt(a) . ntt(s) + ntt(e + m)
// -- the same field elements as the reference's encrypt-then-ntt_forward order (exact arithmetic), one transform per polynomial instead of three.
#include <cmath>
#include <vector>

#include "context.h"
#include "kernels.h"

namespace vpbs {
namespace keygen {
constexpr u64 G = 0x9E3779B97F4A7C15ull;
enum Kind : u64 { S_TO = 1, S_GLWE = 2, BSK_MASK = 3, BSK_NOISE = 4, KSK_MASK = 5, KSK_NOISE = 6, LWE_MASK = 7, LWE_NOISE = 8 };

GL_HD u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
GL_HD u64 tag(u64 kind, u64 a, u64 b, u64 c) { return (kind << 56) | (a << 32) | (b << 16) | c; }
GL_HD u64 stream(u64 seed, u64 t) { return mix64(seed + G * (t + 1)); }
GL_HD u64 draw(u64 stream_key, u64 i) { return mix64(stream_key + G * (i + 1)); }
GL_HD u64 field(u64 u) { return u >= gl::P ? u - gl::P : u; }
GL_HD u64 noise(u64 stream_key, u64 i, u64 m_sigma) {
    u64 s = 0;
#pragma unroll
    for (unsigned k = 0; k < 6; ++k) {
        const u64 d = draw(stream_key, 6 * i + k);
        s += (d & 0xFFFFFFFFull) + (d >> 32);
    }
    const long long t = (long long)s - (6ll << 32);                     // |t| <= 6 2^32
    const __int128 prod = (__int128)t * (__int128)m_sigma;              // m_sigma < 2^62
    const long long e = (long long)(prod >> 32);                        // arithmetic shift = floor
    return e < 0 ? gl::P - (u64)(-e) : (u64)e;
}

// every coefficient of every GLWE of a batch of GGSW encryptions, coefficient domain, straight into the output buffer
// out: [n_ggsw][K][ELL][K][N];  msg: [K][N] (the polynomial GLEV p encrypts before the bit / gadget scaling);  bits: [n_ggsw] or null (= 1)
struct GgswJob {
    u64 seed, m_sigma;
    u64 kind_mask, kind_noise;
    unsigned log_n, K, ELL, n_ggsw;
    u64 scale[16];  // B^(first_limb + l), l < ELL
};
__global__ void __launch_bounds__(256) ggsw_fill_kernel(GgswJob j, const u64* __restrict__ msg, const u64* __restrict__ bits, u64* __restrict__ out) {
    const size_t n = (size_t)1 << j.log_n, total = (size_t)j.n_ggsw * j.K * j.ELL * j.K * n;
    const size_t idx = blockIdx.x * (size_t)256 + threadIdx.x;
    if (idx >= total) return;
    const unsigned i = (unsigned)(idx & (n - 1));
    size_t rest = idx >> j.log_n;
    const unsigned r = (unsigned)(rest % j.K);
    rest /= j.K;
    const unsigned l = (unsigned)(rest % j.ELL);
    rest /= j.ELL;
    const unsigned p = (unsigned)(rest % j.K);
    const unsigned g = (unsigned)(rest / j.K);
    const u64 pl = (u64)p * j.ELL + l;
    if (r + 1 < j.K) {  // mask polynomial r
        out[idx] = field(draw(stream(j.seed, tag(j.kind_mask, g, pl, r)), i));
        return;
    }
    const u64 e = noise(stream(j.seed, tag(j.kind_noise, g, pl, 0)), i, j.m_sigma);
    const u64 bit = bits ? bits[g] : 1;
    const u64 m = bit ? gl::mul(j.scale[l], msg[(size_t)p * n + i]) : 0;
    out[idx] = gl::add(e, m);
}
// body_hat += sum_j a_hat_j . key_hat_j   (all in the NTT domain); one thread per (glwe, coefficient)
__global__ void __launch_bounds__(256) ggsw_body_kernel(unsigned log_n, unsigned K, size_t n_glwe, const u64* __restrict__ key_hat, u64* __restrict__ out) {
    const size_t n = (size_t)1 << log_n, idx = blockIdx.x * (size_t)256 + threadIdx.x;
    if (idx >= n_glwe * n) return;
    const size_t glwe = idx >> log_n, i = idx & (n - 1);
    u64* ct = out + glwe * K * n;
    u64 acc = ct[(size_t)(K - 1) * n + i];
    for (unsigned r = 0; r + 1 < K; ++r) acc = gl::add(acc, gl::mul(ct[(size_t)r * n + i], key_hat[(size_t)r * n + i]));
    ct[(size_t)(K - 1) * n + i] = acc;
}
__global__ void __launch_bounds__(256) pointwise_mac_kernel(size_t n, unsigned terms, const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= n) return;
    u64 acc = 0;
    for (unsigned t = 0; t < terms; ++t) acc = gl::add(acc, gl::mul(a[t * n + i], b[t * n + i]));
    out[i] = acc;
}

inline u64 sigma_to_int(double sigma) {
    const double q = (double)gl::P;  // 18446744069414584320.0, as `F::ORDER as f64`
    return (u64)std::floor(sigma * q + 0.5);
}
inline u64 binary_coeff(u64 seed, u64 kind, unsigned poly, size_t i) { return draw(stream(seed, tag(kind, poly, 0, 0)), i) & 1; }
}  // namespace keygen
}  // namespace vpbs

using vpbs::u64;

namespace {
struct DevBuf {
    vpbs_ctx* c;
    u64* p;
    DevBuf(vpbs_ctx* ctx, size_t words) : c(ctx), p(ctx->alloc_words(words)) {}
    ~DevBuf() {
        (void)vpbs::stream_sync(c->stream);
        c->release(p);
    }
};
bool params_ok(const vpbs_keygen_params* k) {
    if (!k || k->log_N < 1 || k->log_N > 11 || k->K < 2 || k->K > 8 || k->LOGB < 1 || k->LOGB > 32) return false;
    const unsigned nl = (64 + k->LOGB - 1) / k->LOGB;
    if (k->ELL < 1 || k->ELL > nl || k->ELL > 16) return false;
    if (k->n_lwe < 1 || k->n_lwe > (k->K << k->log_N) || k->n_lwe >= (1u << 24)) return false;
    if (!(k->sigma_glwe >= 0.0) || !(k->sigma_lwe >= 0.0) || k->sigma_glwe > 0.2 || k->sigma_lwe > 0.2) return false;
    return true;
}
// Glwe::partial_key(n): polynomial j carries the LWE key coefficients [j N, (j + 1) N) as its leading coefficients, the rest is zero
void host_keys(const vpbs_keygen_params* k, std::vector<u64>& s_to, std::vector<u64>& s_glwe) {
    using namespace vpbs::keygen;
    const size_t n = (size_t)1 << k->log_N;
    s_to.assign((size_t)k->K * n, 0);
    for (size_t x = 0; x < k->n_lwe; ++x) s_to[x] = binary_coeff(k->seed, S_TO, (unsigned)(x / n), x % n);
    s_glwe.assign((size_t)(k->K - 1) * n, 0);
    for (unsigned j = 0; j + 1 < k->K; ++j)
        for (size_t i = 0; i < n; ++i) s_glwe[j * n + i] = binary_coeff(k->seed, S_GLWE, j, i);
}
}  // namespace

extern "C" {

int vpbs_keygen(vpbs_ctx* c, const vpbs_keygen_params* k, uint64_t* s_lwe, uint64_t* s_glwe_out, uint64_t* s_to_out, uint64_t* bsk, uint64_t* ksk,
                int keys_on_device) {
    if (!c || !params_ok(k)) return VPBS_ERR_INVALID;
    try {
        using namespace vpbs;
        using namespace vpbs::keygen;
        VPBS_HIP(hipSetDevice(c->device));
        const unsigned log_n = k->log_N, K = k->K, ELL = k->ELL;
        const size_t n = (size_t)1 << log_n, ggsw_words = (size_t)K * ELL * K * n;
        std::vector<u64> s_to, s_glwe;
        host_keys(k, s_to, s_glwe);
        if (s_lwe) std::copy(s_to.begin(), s_to.begin() + k->n_lwe, s_lwe);   // flatten_partial_key: the leading n coefficients
        if (s_glwe_out) std::copy(s_glwe.begin(), s_glwe.end(), s_glwe_out);
        if (s_to_out) std::copy(s_to.begin(), s_to.end(), s_to_out);
        if (!bsk && !ksk) return VPBS_OK;
        const u64* tab = c->ring_table(log_n);
        const u64 ninv = gl::inv((u64)n);
        // messages of the K GLEVs of a GGSW (before the bit / gadget factor): s_glwe[p] for p < K - 1, the constant 1 for the last
        std::vector<u64> msg((size_t)K * n, 0);
        std::copy(s_glwe.begin(), s_glwe.end(), msg.begin());
        msg[(size_t)(K - 1) * n] = 1;
        DevBuf d_msg(c, msg.size()), d_bits(c, k->n_lwe), d_key(c, (size_t)(K - 1) * n);
        VPBS_HIP(hipMemcpyAsync(d_msg.p, msg.data(), sizeof(u64) * msg.size(), hipMemcpyHostToDevice, c->stream));
        VPBS_HIP(hipMemcpyAsync(d_bits.p, s_to.data(), sizeof(u64) * k->n_lwe, hipMemcpyHostToDevice, c->stream));
        GgswJob job{};
        job.seed = k->seed;
        job.log_n = log_n;
        job.K = K;
        job.ELL = ELL;
        const unsigned first_limb = (64 + k->LOGB - 1) / k->LOGB - ELL;   // ceil_div(F::BITS, LOGB) - ELL (crypto/glev.rs:28)
        for (unsigned l = 0; l < ELL; ++l) job.scale[l] = gl::pow(2, (u64)k->LOGB * (first_limb + l));
        auto run = [&](unsigned n_ggsw, const std::vector<u64>& key, const u64* d_bits_or_null, u64 kind_mask, u64 kind_noise, double sigma, uint64_t* out) {
            job.n_ggsw = n_ggsw;
            job.kind_mask = kind_mask;
            job.kind_noise = kind_noise;
            job.m_sigma = sigma_to_int(sigma);
            const size_t words = (size_t)n_ggsw * ggsw_words;
            u64* d_out = keys_on_device ? out : c->alloc_words(words);
            try {
                VPBS_HIP(hipMemcpyAsync(d_key.p, key.data(), sizeof(u64) * (K - 1) * n, hipMemcpyHostToDevice, c->stream));
                launch_negacyclic(c->stream, d_key.p, tab, K - 1, log_n, false, ninv);
                hipLaunchKernelGGL(ggsw_fill_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream, job, d_msg.p, d_bits_or_null, d_out);
                launch_negacyclic(c->stream, d_out, tab, (unsigned)(words / n), log_n, false, ninv);
                const size_t n_glwe = (size_t)n_ggsw * K * ELL;
                hipLaunchKernelGGL(ggsw_body_kernel, dim3((unsigned)((n_glwe * n + 255) / 256)), dim3(256), 0, c->stream, log_n, K, n_glwe, d_key.p, d_out);
                VPBS_HIP(hipGetLastError());
                if (!keys_on_device) VPBS_HIP(hipMemcpyAsync(out, d_out, sizeof(u64) * words, hipMemcpyDeviceToHost, c->stream));
                VPBS_HIP(vpbs::stream_sync(c->stream));
            } catch (...) {
                (void)vpbs::stream_sync(c->stream);
                if (!keys_on_device) c->release(d_out);
                throw;
            }
            if (!keys_on_device) c->release(d_out);
        };
        // compute_bsk: GGSW i encrypts the constant s_lwe[i] under s_glwe with sigma_glwe
        if (bsk) run(k->n_lwe, s_glwe, d_bits.p, BSK_MASK, BSK_NOISE, k->sigma_glwe, bsk);
        // compute_ksk(s_to, s_glwe, sigma_lwe): GLEV i encrypts s_glwe[i] (i < K - 1) / the constant 1 under s_to
        if (ksk) run(1, s_to, nullptr, KSK_MASK, KSK_NOISE, k->sigma_lwe, ksk);
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        c->err = e.what;
        return e.status;
    }
}

int vpbs_lwe_encrypt(const vpbs_keygen_params* k, const uint64_t* s_lwe, uint64_t message, uint64_t nonce, uint64_t* ct) {
    if (!params_ok(k) || !s_lwe || !ct || message >= gl::P || nonce >= (1ull << 24)) return VPBS_ERR_INVALID;
    using namespace vpbs::keygen;
    // crypto/lwe.rs:55-64: mask uniform, body = <s, mask> + m + e
    const u64 km = stream(k->seed, tag(LWE_MASK, nonce, 0, 0)), ke = stream(k->seed, tag(LWE_NOISE, nonce, 0, 0));
    u64 body = gl::add(message, noise(ke, 0, sigma_to_int(k->sigma_lwe)));
    for (unsigned i = 0; i < k->n_lwe; ++i) {
        ct[i] = field(draw(km, i));
        if (s_lwe[i]) body = gl::add(body, gl::mul(ct[i], s_lwe[i] % gl::P));
    }
    ct[k->n_lwe] = body;
    return VPBS_OK;
}

int vpbs_testv(unsigned log_N, unsigned p, uint64_t* testv, uint64_t* delta_out) {
    if (log_N < 1 || log_N > 16 || p < 1 || (p & (p - 1)) || p > (1u << log_N)) return VPBS_ERR_INVALID;
    const size_t n = (size_t)1 << log_N;
    unsigned log2p = 0;
    while ((1u << log2p) < 2 * p) ++log2p;
    const u64 delta = gl::P >> log2p;  // get_delta(2 p) = F::order() >> log2_ceil(2 p)   (main.rs:47)
    if (delta_out) *delta_out = delta;
    if (!testv) return VPBS_OK;
    // get_testv: blocks of N / p equal coefficients i delta, then Poly::left_shift(block / 2): c[i] <- c[i + s], wrapped terms negated
    const size_t block = n / p, s = block / 2;
    auto coeff = [&](size_t i) { return gl::mul((u64)(i / block), delta); };
    for (size_t i = 0; i < n; ++i) testv[i] = i + s < n ? coeff(i + s) : gl::neg(coeff(i + s - n));
    return VPBS_OK;
}

int vpbs_glwe_decrypt(vpbs_ctx* c, unsigned log_N, unsigned K, const uint64_t* s, const uint64_t* ct, uint64_t* m_out) {
    if (!c || !s || !ct || !m_out || log_N < 1 || log_N > 11 || K < 2 || K > 8) return VPBS_ERR_INVALID;
    try {
        using namespace vpbs;
        VPBS_HIP(hipSetDevice(c->device));
        const size_t n = (size_t)1 << log_N;
        const u64* tab = c->ring_table(log_N);
        const u64 ninv = gl::inv((u64)n);
        // crypto/glwe.rs:60-63: body - sum_j a_j s_j  (negacyclic products through the NTT)
        DevBuf a(c, (size_t)(K - 1) * n), k(c, (size_t)(K - 1) * n), prod(c, n);
        VPBS_HIP(hipMemcpyAsync(a.p, ct, sizeof(u64) * (K - 1) * n, hipMemcpyHostToDevice, c->stream));
        VPBS_HIP(hipMemcpyAsync(k.p, s, sizeof(u64) * (K - 1) * n, hipMemcpyHostToDevice, c->stream));
        launch_negacyclic(c->stream, a.p, tab, K - 1, log_N, false, ninv);
        launch_negacyclic(c->stream, k.p, tab, K - 1, log_N, false, ninv);
        hipLaunchKernelGGL(keygen::pointwise_mac_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, K - 1, a.p, k.p, prod.p);
        launch_negacyclic(c->stream, prod.p, tab + n, 1, log_N, true, ninv);
        std::vector<u64> mask(n);
        VPBS_HIP(hipMemcpyAsync(mask.data(), prod.p, sizeof(u64) * n, hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
        for (size_t i = 0; i < n; ++i) m_out[i] = gl::sub(ct[(size_t)(K - 1) * n + i], mask[i]);
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        c->err = e.what;
        return e.status;
    }
}

}  // extern "C"
