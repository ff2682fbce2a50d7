/* ORACLE (test infrastructure).  Cyclic FFT over Goldilocks as used by the plonky2 prover:
 * plonky2_field 0.2.0 fft.rs (fft_root_table / fft_classic / ifft_with_options) and polynomial/mod.rs
 * (PolynomialValues::ifft, PolynomialCoeffs::lde, coset_fft_with_options) -- SURVEY.md 8a row a3, Appendix A.4.
 * Conventions: w_k = primitive_root_of_unity(k) = POWER_OF_TWO_GENERATOR^(2^(32-k)); natural order in and out;
 * ifft scales by n^-1; coset_fft multiplies coefficient i by shift^i first.
 * Also the reference's own negacyclic NTT: /root/reference/src/vtfhe/crypto/poly.rs:9-64, parameter tables per
 * /root/reference/src/ntt/gen_param_file.sage:3-10,97-120 (pinned by TESTG/TESTGHAT, params_{N}.rs:11,13). */
#include "vpbs_oracle.h"
#include <stdlib.h>
#include <string.h>

static void fft_in_place(u64* a, unsigned log_n, u64 root) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) {
        size_t j = bitrev(i, log_n);
        if (i < j) { u64 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    u64* tw = (u64*)malloc(sizeof(u64) * (n / 2 + 1));
    tw[0] = 1;
    for (size_t i = 1; i < n / 2; ++i) tw[i] = gl_mul(tw[i - 1], root);
    for (unsigned s = 1; s <= log_n; ++s) {
        size_t m = (size_t)1 << s, half = m >> 1, step = n >> s;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; ++j) {
                u64 t = gl_mul(tw[j * step], a[k + j + half]);
                u64 u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + half] = gl_sub(u, t);
            }
    }
    free(tw);
}

void orc_fft(u64* a, unsigned log_n) { fft_in_place(a, log_n, gl_root_of_unity(log_n)); }

void orc_ifft(u64* a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    fft_in_place(a, log_n, gl_inv(gl_root_of_unity(log_n)));
    u64 ninv = gl_inv((u64)n);
    for (size_t i = 0; i < n; ++i) a[i] = gl_mul(a[i], ninv);
}

void orc_coset_lde(const u64* coeffs, unsigned log_n, unsigned rate_bits, u64 shift, u64* out) {
    size_t n = (size_t)1 << log_n, big = n << rate_bits;
    u64 pw = 1;
    for (size_t i = 0; i < n; ++i) { out[i] = gl_mul(coeffs[i], pw); pw = gl_mul(pw, shift); }
    memset(out + n, 0, (big - n) * sizeof(u64));
    orc_fft(out, log_n + rate_bits);
}

/* ---- negacyclic transform of the reference ---- */
void orc_negacyclic_params(unsigned log_n, u64* roots, u64* invroots, u64* ninv) {
    /* psi = 7^((p-1)/2N); ROOTS[j] = psi^brev(j), INVROOTS[j] = psi^-brev(j)  (gen_param_file.sage) */
    size_t n = (size_t)1 << log_n;
    u64 psi = gl_exp(GL_GENERATOR, (GL_P - 1) / (2 * n));
    u64 psi_inv = gl_inv(psi);
    for (size_t j = 0; j < n; ++j) {
        size_t e = bitrev(j, log_n);
        roots[j] = gl_exp(psi, e);
        invroots[j] = gl_exp(psi_inv, e);
    }
    *ninv = gl_inv((u64)n);
}

/* poly.rs:9-34 ntt_fw_update / ntt_forward (Cooley-Tukey, natural in, bit-reversed out) */
void orc_negacyclic_forward(u64* a, unsigned log_n, const u64* roots) {
    size_t n = (size_t)1 << log_n;
    for (size_t m = 1; m < n; m <<= 1) {
        size_t t = n / (2 * m);
        for (size_t i = 0; i < m; ++i) {
            size_t j1 = 2 * i * t, j2 = j1 + t;
            u64 s = roots[m + i];
            for (size_t j = j1; j < j2; ++j) {
                u64 u = a[j], v = gl_mul(a[j + t], s);
                a[j] = gl_add(u, v);
                a[j + t] = gl_sub(u, v);
            }
        }
    }
}

/* poly.rs:36-64 ntt_bw_update / ntt_backward (Gentleman-Sande, then * N^-1) */
void orc_negacyclic_backward(u64* a, unsigned log_n, const u64* invroots, u64 ninv) {
    size_t n = (size_t)1 << log_n;
    for (size_t m = n >> 1; m >= 1; m >>= 1) {
        size_t t = n / (2 * m), j1 = 0;
        for (size_t i = 0; i < m; ++i) {
            size_t j2 = j1 + t;
            u64 s = invroots[m + i];
            for (size_t j = j1; j < j2; ++j) {
                u64 u = a[j], v = a[j + t];
                a[j] = gl_add(u, v);
                a[j + t] = gl_mul(gl_sub(u, v), s);
            }
            j1 += 2 * t;
        }
    }
    for (size_t i = 0; i < n; ++i) a[i] = gl_mul(a[i], ninv);
}
