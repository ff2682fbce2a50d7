/* ORACLE (test infrastructure).  Permutation-argument partial products and Z polynomials.
 * Restates plonky2 0.2.0 plonk/prover.rs `all_wires_permutation_partial_products` /
 * `wires_permutation_partial_products_and_zs`, plonk/plonk_common.rs-adjacent helpers `quotient_chunk_products` and
 * `partial_products_and_z_gx` (plonk/vanishing_poly.rs / util/partial_products.rs), and
 * plonk/permutation_argument.rs `get_unique_coset_shifts` (k_is[j] = 7^j) -- SURVEY.md 8a row a12, Appendix A.9;
 * reached from prove() at /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364.  Sequential, literal restatement
 * (per-element batch inverse, running z_x).  parity unpinned against real plonky2 output. */
#include "vpbs_oracle.h"
#include <stdlib.h>

int orc_partial_products(const u64* wires, const u64* sigmas, size_t n_routed, unsigned log_n, const u64* betas,
                         const u64* gammas, size_t num_challenges, size_t max_degree, u64* out) {
    size_t n = (size_t)1 << log_n;
    size_t n_chunks = (n_routed + max_degree - 1) / max_degree; /* = num_prods + 1 */
    size_t num_prods = n_chunks - 1;
    u64* k_is = (u64*)malloc(sizeof(u64) * n_routed);
    u64* den = (u64*)malloc(sizeof(u64) * n_routed);
    u64* pre = (u64*)malloc(sizeof(u64) * n_routed);
    u64* q = (u64*)malloc(sizeof(u64) * n_routed);
    k_is[0] = 1;
    for (size_t j = 1; j < n_routed; ++j) k_is[j] = gl_mul(k_is[j - 1], GL_GENERATOR);
    u64 w = gl_root_of_unity(log_n);
    int rc = 0;
    for (size_t c = 0; c < num_challenges && rc == 0; ++c) {
        u64 beta = betas[c], gamma = gammas[c];
        u64* z_col = out + c * n;
        u64* pp_base = out + (num_challenges + c * num_prods) * n;
        u64 z_x = 1, x = 1;
        for (size_t i = 0; i < n; ++i) {
            /* denominators and their batch inverse (F::batch_multiplicative_inverse) */
            for (size_t j = 0; j < n_routed; ++j)
                den[j] = gl_add(gl_add(wires[j * n + i], gl_mul(beta, sigmas[j * n + i])), gamma);
            u64 acc = 1;
            for (size_t j = 0; j < n_routed; ++j) { pre[j] = acc; acc = gl_mul(acc, den[j]); }
            if (acc == 0) { rc = -1; break; } /* plonky2 would panic on a zero denominator */
            u64 inv = gl_inv(acc);
            for (size_t j = n_routed; j-- > 0;) { u64 dj = den[j]; den[j] = gl_mul(inv, pre[j]); inv = gl_mul(inv, dj); }
            for (size_t j = 0; j < n_routed; ++j) {
                u64 num = gl_add(gl_add(wires[j * n + i], gl_mul(beta, gl_mul(k_is[j], x))), gamma);
                q[j] = gl_mul(num, den[j]);
            }
            /* quotient_chunk_products + partial_products_and_z_gx, last entry swapped with Z(x) */
            u64 run = z_x;
            for (size_t k = 0; k < n_chunks; ++k) {
                u64 prod = 1;
                for (size_t j = k * max_degree; j < (k + 1) * max_degree && j < n_routed; ++j) prod = gl_mul(prod, q[j]);
                run = gl_mul(run, prod);
                if (k < num_prods) pp_base[k * n + i] = run;
            }
            z_col[i] = z_x; /* Z(x) */
            z_x = run;      /* Z(g x) */
            x = gl_mul(x, w);
        }
    }
    free(k_is); free(den); free(pre); free(q);
    return rc;
}
