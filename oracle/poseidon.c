/* ORACLE (test infrastructure).  Poseidon over Goldilocks, width 12 -- restates plonky2 0.2.0
 * hash/poseidon.rs (`Poseidon::poseidon`, naive round structure: constant layer, S-box x^7, MDS layer) with the
 * constants of hash/poseidon_goldilocks.rs (MDS_MATRIX_CIRC / MDS_MATRIX_DIAG, ALL_ROUND_CONSTANTS) and the
 * sponge helpers of hash/hashing.rs (hash_n_to_m_no_pad: overwrite mode, rate 8) -- SURVEY.md Appendix A.2.
 * Used natively by the reference at /root/reference/src/vtfhe/ivc_based_vpbs.rs:73.
 * Pinned by the upstream KATs in tests/golden/poseidon_kat.json. */
#include "vpbs_oracle.h"
#include "poseidon_constants.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static inline u64 sbox7(u64 x) {
    u64 x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x2, x);
    return gl_mul(x3, x4);
}

void orc_poseidon(u64 s[12]) {
    for (int r = 0; r < 30; ++r) {
        u64 d[24]; /* state after the constant + S-box layers, stored twice so the circulant index needs no modulo */
        for (int i = 0; i < 12; ++i) d[i] = gl_add(s[i], POSEIDON_RC[12 * r + i]);
        if (r < 4 || r >= 26) { for (int i = 0; i < 12; ++i) d[i] = sbox7(d[i]); }
        else d[0] = sbox7(d[0]);
        memcpy(d + 12, d, 12 * sizeof(u64));
        for (int row = 0; row < 12; ++row) {
            u128 acc = 0; /* 12 terms of < 2^64 * 2^6 : fits easily */
            for (int i = 0; i < 12; ++i) acc += (u128)d[i + row] * MDS_CIRC[i];
            acc += (u128)d[row] * MDS_DIAG[row];
            s[row] = gl_reduce128((u64)acc, (u64)(acc >> 64));
        }
    }
}

void orc_poseidon_batch(u64* states, size_t n) {
    if (n >= 8 && orc_poseidon_x8_available()) {
        orc_poseidon_batch_x8(states, n);
        return;
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) orc_poseidon(states + 12 * i);
}

void orc_hash_no_pad(const u64* in, size_t n, u64 out[4]) {
    u64 s[12] = {0};
    for (size_t off = 0; off < n; off += 8) {
        size_t len = n - off < 8 ? n - off : 8;
        memcpy(s, in + off, len * sizeof(u64)); /* overwrite mode */
        orc_poseidon(s);
    }
    memcpy(out, s, 4 * sizeof(u64));
}

/* Hasher::hash_pad (plonk/config.rs): padded_input.push(ONE); while (len + 1) % RATE != 0 push(ZERO); push(ONE); hash_no_pad */
void orc_hash_pad(const u64* in, size_t n, u64 out[4]) {
    size_t len = n + 1;
    while ((len + 1) % 8 != 0) ++len;
    ++len;
    u64* buf = (u64*)calloc(len, sizeof(u64));
    if (n) memcpy(buf, in, n * sizeof(u64));
    buf[n] = 1;
    buf[len - 1] = 1;
    orc_hash_no_pad(buf, len, out);
    free(buf);
}

void orc_compat_default(orc_compat* out) {
    out->fri_mul_final_by_x = 0;
    out->bytes_pi_len_prefix = 1;
    out->digest_domain_separator = 1;
    out->pow_smallest_nonce = 1;
}

/* plonk/circuit_builder.rs build(): circuit_digest_parts = [constants_sigmas_cap.flatten(), hash_pad(domain_separator = []).to_vec(),
 * [degree_bits]]; circuit_digest = hash_no_pad(concat).  The reference sets no domain separator (ivc_based_vpbs.rs:190-276). */
void orc_circuit_digest(const orc_compat* compat, const u64* cap, size_t cap_words, unsigned degree_bits, u64 out[4]) {
    orc_compat k;
    orc_compat_default(&k);
    if (compat) k = *compat;
    u64* buf = (u64*)calloc(cap_words + 5, sizeof(u64));
    size_t len = cap_words;
    memcpy(buf, cap, cap_words * sizeof(u64));
    if (k.digest_domain_separator) {
        orc_hash_pad(NULL, 0, buf + len);
        len += 4;
    }
    buf[len++] = degree_bits;
    orc_hash_no_pad(buf, len, out);
    free(buf);
}

void orc_hash_or_noop(const u64* in, size_t n, u64 out[4]) {
    if (n <= 4) {
        memset(out, 0, 4 * sizeof(u64));
        memcpy(out, in, n * sizeof(u64));
    } else orc_hash_no_pad(in, n, out);
}

void orc_two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
    u64 s[12] = {0};
    memcpy(s, l, 4 * sizeof(u64));
    memcpy(s + 4, r, 4 * sizeof(u64));
    orc_poseidon(s);
    memcpy(out, s, 4 * sizeof(u64));
}

/* h <- hash_no_pad(h || data_k), h_0 = 0^4 : /root/reference/src/vtfhe/ivc_based_vpbs.rs:64-78 */
void orc_hash_chain(const u64* data, size_t n_items, size_t item_len, u64 out[4]) {
    u64 h[4] = {0};
    u64 buf[4 + 65536];
    for (size_t k = 0; k < n_items; ++k) {
        if (item_len > 65536) return;
        memcpy(buf, h, sizeof h);
        memcpy(buf + 4, data + k * item_len, item_len * sizeof(u64));
        orc_hash_no_pad(buf, 4 + item_len, h);
    }
    memcpy(out, h, sizeof h);
}

u64 orc_gl_add(u64 a, u64 b) { return gl_add(a, b); }
u64 orc_gl_sub(u64 a, u64 b) { return gl_sub(a, b); }
u64 orc_gl_mul(u64 a, u64 b) { return gl_mul(a, b); }
u64 orc_gl_inv(u64 a) { return gl_inv(a); }
u64 orc_gl_exp(u64 a, u64 e) { return gl_exp(a, e); }
u64 orc_gl_root_of_unity(unsigned k) { return gl_root_of_unity(k); }
void orc_ext_mul(const u64 a[2], const u64 b[2], u64 out[2]) {
    ext2 r = ext_mul(ext_make(a[0], a[1]), ext_make(b[0], b[1]));
    out[0] = r.c[0]; out[1] = r.c[1];
}
void orc_ext_inv(const u64 a[2], u64 out[2]) {
    ext2 r = ext_inv(ext_make(a[0], a[1]));
    out[0] = r.c[0]; out[1] = r.c[1];
}

/* number of OpenMP threads the oracle uses (the test harness sets it to the CPU quota of the container: a 256-CPU host with a
 * 16-core cgroup quota runs 256 threads far slower than 16) */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
