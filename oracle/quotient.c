/* ORACLE (test infrastructure).  Quotient polynomials for the permutation argument + the verifier's vanishing check.
 * Restates plonky2 0.2.0 plonk/prover.rs `compute_quotient_polys`, plonk/vanishing_poly.rs
 * `eval_vanishing_poly_base_batch` / `eval_vanishing_poly` (permutation part: L_0 (Z - 1) terms and
 * `check_partial_products`), plonk/plonk_common.rs `ZeroPolyOnCoset` (Z_H on the coset, eval_l_0) and
 * `reduce_with_powers_multi`, plonk/verifier.rs (the check vanishing(zeta) == Z_H(zeta) * reduce(chunks, zeta^n)) --
 * SURVEY.md 8a row a13, Appendix A.9; reached from prove() at /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364
 * and cd.verify() at :446.  Gate-constraint terms are an input (alpha-folded per challenge), produced by gates.c.
 * parity unpinned against real plonky2; prover and verifier sides are
 * checked against each other (a valid copy-constraint witness must verify, an invalid one must not). */
#include "vpbs_oracle.h"
#include <stdlib.h>
#include <string.h>

int orc_quotient_permutation(const u64* wires_c, const u64* sigmas_c, const u64* zs_pp_c, size_t n_routed, unsigned log_n,
                             const u64* betas, const u64* gammas, const u64* alphas, size_t nc, size_t max_degree,
                             const u64* gate_terms, u64* out) {
    const unsigned rate_bits = 3;
    size_t n = (size_t)1 << log_n, big = n << rate_bits;
    size_t n_chunks = (n_routed + max_degree - 1) / max_degree, num_prods = n_chunks - 1;
    size_t n_zs = nc * n_chunks;
    /* LDEs in natural coset order */
    u64* W = (u64*)malloc(sizeof(u64) * n_routed * big);
    u64* S = (u64*)malloc(sizeof(u64) * n_routed * big);
    u64* Zp = (u64*)malloc(sizeof(u64) * n_zs * big);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t j = 0; j < 2 * n_routed + n_zs; ++j) {
        if (j < n_routed) orc_coset_lde(wires_c + j * n, log_n, rate_bits, GL_GENERATOR, W + j * big);
        else if (j < 2 * n_routed) orc_coset_lde(sigmas_c + (j - n_routed) * n, log_n, rate_bits, GL_GENERATOR, S + (j - n_routed) * big);
        else orc_coset_lde(zs_pp_c + (j - 2 * n_routed) * n, log_n, rate_bits, GL_GENERATOR, Zp + (j - 2 * n_routed) * big);
    }
    u64* k_is = (u64*)malloc(sizeof(u64) * n_routed);
    k_is[0] = 1;
    for (size_t j = 1; j < n_routed; ++j) k_is[j] = gl_mul(k_is[j - 1], GL_GENERATOR);
    u64 w_big = gl_root_of_unity(log_n + rate_bits);
    /* Z_H on the coset takes 2^rate_bits values: (7 w^t)^n - 1 = 7^n w_8^(t mod 8) - 1 */
    u64 zh_inv[8];
    u64 seven_n = gl_exp(GL_GENERATOR, n), w8 = gl_root_of_unity(rate_bits);
    for (size_t r = 0; r < 8; ++r) zh_inv[r] = gl_inv(gl_sub(gl_mul(seven_n, gl_exp(w8, r)), 1));
    u64 n_field = gl_from_u64((u64)n);
    size_t n_terms = nc + nc * n_chunks;
    u64* Q = (u64*)malloc(sizeof(u64) * nc * big);
    int rc = 0;
#pragma omp parallel for schedule(static)
    for (size_t t = 0; t < big; ++t) {
        u64 terms[64];
        u64 x = gl_mul(GL_GENERATOR, gl_exp(w_big, t));
        u64 zh = gl_sub(gl_mul(seven_n, gl_exp(w8, t & 7)), 1);
        u64 l0 = gl_mul(zh, gl_inv(gl_mul(n_field, gl_sub(x, 1)))); /* eval_l_0 */
        size_t t_next = (t + ((size_t)1 << rate_bits)) & (big - 1);    /* g x: next row of the trace */
        for (size_t c = 0; c < nc; ++c) {
            const u64* Zc = Zp + c * big;
            const u64* PPc = Zp + (nc + c * num_prods) * big;
            terms[c] = gl_mul(l0, gl_sub(Zc[t], 1));
            u64 s_id_beta = gl_mul(betas[c], x);
            for (size_t k = 0; k < n_chunks; ++k) {
                u64 num = 1, den = 1;
                for (size_t j = k * max_degree; j < (k + 1) * max_degree && j < n_routed; ++j) {
                    u64 wv = W[j * big + t];
                    num = gl_mul(num, gl_add(gl_add(wv, gl_mul(s_id_beta, k_is[j])), gammas[c]));
                    den = gl_mul(den, gl_add(gl_add(wv, gl_mul(betas[c], S[j * big + t])), gammas[c]));
                }
                u64 prev = k == 0 ? Zc[t] : PPc[(k - 1) * big + t];
                u64 next = k == num_prods ? Zc[t_next] : PPc[k * big + t];
                terms[nc + c * n_chunks + k] = gl_sub(gl_mul(prev, num), gl_mul(next, den));
            }
        }
        for (size_t a = 0; a < nc; ++a) { /* reduce_with_powers(terms, alpha_a) (+ the folded gate terms behind them) */
            u64 acc = gate_terms ? gate_terms[a * big + t] : 0;
            for (size_t i = n_terms; i-- > 0;) acc = gl_add(gl_mul(acc, alphas[a]), terms[i]);
            Q[a * big + t] = gl_mul(acc, zh_inv[t & 7]);
        }
    }
    /* coset_ifft(7): ifft then coefficient i times 7^-i; chunks of n */
    u64 inv7 = gl_inv(GL_GENERATOR);
    for (size_t a = 0; a < nc; ++a) {
        u64* q = Q + a * big;
        orc_ifft(q, log_n + rate_bits);
        u64 pw = 1;
        for (size_t i = 0; i < big; ++i) { out[a * big + i] = gl_mul(q[i], pw); pw = gl_mul(pw, inv7); }
    }
    free(W); free(S); free(Zp); free(k_is); free(Q);
    return rc;
}

int orc_check_vanishing_at_zeta(const u64* wires_z, const u64* sigmas_z, const u64* zs_z, const u64* zs_next_z,
                                const u64* pps_z, const u64* quotient_z, size_t n_routed, unsigned log_n, const u64* betas,
                                const u64* gammas, const u64* alphas, size_t nc, size_t max_degree, const u64 zeta_w[2],
                                const u64* gate_terms_zeta) {
    size_t n = (size_t)1 << log_n;
    size_t n_chunks = (n_routed + max_degree - 1) / max_degree, num_prods = n_chunks - 1;
    ext2 zeta = ext_make(zeta_w[0], zeta_w[1]);
    ext2 zeta_n = zeta;
    for (unsigned i = 0; i < log_n; ++i) zeta_n = ext_mul(zeta_n, zeta_n);
    ext2 one = ext_from_base(1);
    ext2 z_h = ext_sub(zeta_n, one);
    ext2 l0 = ext_mul(z_h, ext_inv(ext_scalar_mul(ext_sub(zeta, one), gl_from_u64((u64)n))));
#define EXT_AT(p, i) ext_make((p)[2 * (i)], (p)[2 * (i) + 1])
    ext2 terms[64];
    u64 k = 1;
    u64 k_is[256];
    for (size_t j = 0; j < n_routed; ++j) { k_is[j] = k; k = gl_mul(k, GL_GENERATOR); }
    for (size_t c = 0; c < nc; ++c) {
        terms[c] = ext_mul(l0, ext_sub(EXT_AT(zs_z, c), one));
        for (size_t kk = 0; kk < n_chunks; ++kk) {
            ext2 num = one, den = one;
            for (size_t j = kk * max_degree; j < (kk + 1) * max_degree && j < n_routed; ++j) {
                ext2 wv = EXT_AT(wires_z, j);
                ext2 g = ext_from_base(gammas[c]);
                num = ext_mul(num, ext_add(ext_add(wv, ext_scalar_mul(zeta, gl_mul(betas[c], k_is[j]))), g));
                den = ext_mul(den, ext_add(ext_add(wv, ext_scalar_mul(EXT_AT(sigmas_z, j), betas[c])), g));
            }
            ext2 prev = kk == 0 ? EXT_AT(zs_z, c) : EXT_AT(pps_z, c * num_prods + kk - 1);
            ext2 next = kk == num_prods ? EXT_AT(zs_next_z, c) : EXT_AT(pps_z, c * num_prods + kk);
            terms[nc + c * n_chunks + kk] = ext_sub(ext_mul(prev, num), ext_mul(next, den));
        }
    }
    size_t n_terms = nc + nc * n_chunks;
    for (size_t a = 0; a < nc; ++a) {
        ext2 acc = gate_terms_zeta ? EXT_AT(gate_terms_zeta, a) : ext_from_base(0);
        for (size_t i = n_terms; i-- > 0;) acc = ext_add(ext_scalar_mul(acc, alphas[a]), terms[i]);
        /* z_h_zeta * reduce_with_powers(chunks, zeta^n) */
        ext2 q = ext_from_base(0);
        for (size_t m = 8; m-- > 0;) q = ext_add(ext_mul(q, zeta_n), EXT_AT(quotient_z, a * 8 + m));
        if (!ext_eq(acc, ext_mul(z_h, q))) return 0;
    }
#undef EXT_AT
    return 1;
}
