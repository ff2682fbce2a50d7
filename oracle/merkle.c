/* ORACLE (test infrastructure).  Merkle tree with cap + PolynomialBatch + Challenger.
 * Restates plonky2 0.2.0 hash/merkle_tree.rs (MerkleTree::new / get / prove, MerkleCap), hash/merkle_proofs.rs
 * (verify_merkle_proof_to_cap), fri/oracle.rs (PolynomialBatch::from_values / from_coeffs / get_lde_values),
 * plonk/proof.rs (OpeningSet::new evaluation), iop/challenger.rs -- SURVEY.md 8a rows a4-a9, Appendix A.4/A.5.
 * digest(leaf) = hash_or_noop(leaf); parent = two_to_one(left, right); cap[i] = root over leaves
 * [i*L/2^h, (i+1)*L/2^h); proof = siblings from the leaf level up to (excluding) the cap level. */
#include "vpbs_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
static double orc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define ORC_TRACE(label, t0) do { if (getenv("ORC_TRACE")) fprintf(stderr, "[orc] %-22s %.3f s\n", label, orc_now() - (t0)); } while (0)

struct orc_merkle {
    size_t n_leaves, leaf_len;
    unsigned log_leaves, cap_height;
    u64* leaves;   /* [n_leaves][leaf_len] (owned copy) */
    u64** levels;  /* levels[0] = leaf digests [n][4]; levels[k] has n >> k nodes; up to level log_leaves - cap_height */
};

orc_merkle* orc_merkle_new(const u64* leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height) {
    unsigned log_leaves = 0;
    while (((size_t)1 << log_leaves) < n_leaves) ++log_leaves;
    if (((size_t)1 << log_leaves) != n_leaves || cap_height > log_leaves) return NULL;
    orc_merkle* t = (orc_merkle*)calloc(1, sizeof *t);
    t->n_leaves = n_leaves; t->leaf_len = leaf_len; t->log_leaves = log_leaves; t->cap_height = cap_height;
    t->leaves = (u64*)malloc(sizeof(u64) * n_leaves * leaf_len);
    double t0 = orc_now();
    memcpy(t->leaves, leaves, sizeof(u64) * n_leaves * leaf_len);
    ORC_TRACE("  leaves memcpy", t0); t0 = orc_now();
    unsigned n_levels = log_leaves - cap_height + 1;
    t->levels = (u64**)calloc(n_levels, sizeof(u64*));
    t->levels[0] = (u64*)malloc(sizeof(u64) * 4 * n_leaves);
    const int x8 = orc_poseidon_x8_available();   /* eight leaves / nodes per permutation on AVX-512 lanes, same digests */
    if (x8) {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n_leaves; i += 8)
            orc_hash_rows_x8(t->leaves + i * leaf_len, leaf_len, leaf_len, n_leaves - i < 8 ? n_leaves - i : 8, t->levels[0] + 4 * i);
    } else {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n_leaves; ++i) orc_hash_or_noop(t->leaves + i * leaf_len, leaf_len, t->levels[0] + 4 * i);
    }
    ORC_TRACE("  leaf hashing", t0); t0 = orc_now();
    for (unsigned k = 1; k < n_levels; ++k) {
        size_t cnt = n_leaves >> k;
        t->levels[k] = (u64*)malloc(sizeof(u64) * 4 * cnt);
        if (x8 && cnt >= 8) {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < cnt; i += 8) orc_two_to_one_x8(t->levels[k - 1] + 8 * i, cnt - i < 8 ? cnt - i : 8, t->levels[k] + 4 * i);
            continue;
        }
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < cnt; ++i)
            orc_two_to_one(t->levels[k - 1] + 8 * i, t->levels[k - 1] + 8 * i + 4, t->levels[k] + 4 * i);
    }
    ORC_TRACE("  levels", t0);
    return t;
}

void orc_merkle_free(orc_merkle* t) {
    if (!t) return;
    unsigned n_levels = t->log_leaves - t->cap_height + 1;
    for (unsigned k = 0; k < n_levels; ++k) free(t->levels[k]);
    free(t->levels); free(t->leaves); free(t);
}

void orc_merkle_cap(const orc_merkle* t, u64* cap_out) {
    memcpy(cap_out, t->levels[t->log_leaves - t->cap_height], sizeof(u64) * 4 * ((size_t)1 << t->cap_height));
}
size_t orc_merkle_proof_len(const orc_merkle* t) { return t->log_leaves - t->cap_height; }
void orc_merkle_leaf(const orc_merkle* t, size_t idx, u64* leaf_out) {
    memcpy(leaf_out, t->leaves + idx * t->leaf_len, sizeof(u64) * t->leaf_len);
}
void orc_merkle_prove(const orc_merkle* t, size_t idx, u64* sib) {
    size_t len = orc_merkle_proof_len(t);
    for (size_t k = 0; k < len; ++k) {
        memcpy(sib + 4 * k, t->levels[k] + 4 * ((idx >> k) ^ 1), 4 * sizeof(u64));
    }
}
int orc_merkle_verify(const u64* leaf, size_t leaf_len, size_t idx, const u64* cap, unsigned cap_height,
                      const u64* sib, size_t n_sib) {
    u64 cur[4];
    (void)cap_height;
    orc_hash_or_noop(leaf, leaf_len, cur);
    for (size_t k = 0; k < n_sib; ++k) {
        u64 nxt[4];
        if (idx & 1) orc_two_to_one(sib + 4 * k, cur, nxt); else orc_two_to_one(cur, sib + 4 * k, nxt);
        memcpy(cur, nxt, sizeof cur);
        idx >>= 1;
    }
    return memcmp(cur, cap + 4 * idx, sizeof cur) == 0;
}

/* ---------------- PolynomialBatch ---------------- */
struct orc_batch {
    size_t ncols;
    unsigned log_n, rate_bits;
    u64* coeffs;       /* [ncols][n] */
    orc_merkle* tree;  /* leaves [n<<rate][ncols] in reverse_index_bits order */
};

orc_batch* orc_batch_from_coeffs(const u64* coeffs, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height) {
    size_t n = (size_t)1 << log_n, big = n << rate_bits;
    unsigned log_big = log_n + rate_bits;
    orc_batch* b = (orc_batch*)calloc(1, sizeof *b);
    b->ncols = ncols; b->log_n = log_n; b->rate_bits = rate_bits;
    b->coeffs = (u64*)malloc(sizeof(u64) * ncols * n);
    memcpy(b->coeffs, coeffs, sizeof(u64) * ncols * n);
    u64* leaves = (u64*)malloc(sizeof(u64) * big * ncols);
    u64* lde = (u64*)malloc(sizeof(u64) * big * ncols); /* column-major [ncols][big], natural order */
    double t0 = orc_now();
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t c = 0; c < ncols; ++c) orc_coset_lde(b->coeffs + c * n, log_n, rate_bits, GL_GENERATOR, lde + c * big);
    ORC_TRACE("lde", t0); t0 = orc_now();
    /* transpose + reverse_index_bits_in_place: leaves[j][c] = lde_c[brev(j)]; parallel over leaf rows so that no two
     * threads write the same cache line */
#pragma omp parallel for schedule(static)
    for (size_t s0 = 0; s0 < big; s0 += 16) { /* 16 consecutive natural indices x 8 columns: 64-byte row segments */
        size_t hi = s0 + 16 < big ? s0 + 16 : big;
        size_t rows[16];
        for (size_t t = s0; t < hi; ++t) rows[t - s0] = bitrev(t, log_big);
        for (size_t c0 = 0; c0 < ncols; c0 += 8) {
            size_t c1 = c0 + 8 < ncols ? c0 + 8 : ncols;
            for (size_t t = s0; t < hi; ++t)
                for (size_t c = c0; c < c1; ++c) leaves[rows[t - s0] * ncols + c] = lde[c * big + t];
        }
    }
    ORC_TRACE("transpose", t0); t0 = orc_now();
    free(lde);
    b->tree = orc_merkle_new(leaves, big, ncols, cap_height);
    ORC_TRACE("merkle_new", t0);
    free(leaves);
    if (!b->tree) { free(b->coeffs); free(b); return NULL; }
    return b;
}

orc_batch* orc_batch_from_values(const u64* values, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height) {
    size_t n = (size_t)1 << log_n;
    u64* coeffs = (u64*)malloc(sizeof(u64) * ncols * n);
    memcpy(coeffs, values, sizeof(u64) * ncols * n);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t c = 0; c < ncols; ++c) orc_ifft(coeffs + c * n, log_n);
    orc_batch* b = orc_batch_from_coeffs(coeffs, ncols, log_n, rate_bits, cap_height);
    free(coeffs);
    return b;
}

void orc_batch_free(orc_batch* b) { if (b) { orc_merkle_free(b->tree); free(b->coeffs); free(b); } }
void orc_batch_cap(const orc_batch* b, u64* cap_out) { orc_merkle_cap(b->tree, cap_out); }
const u64* orc_batch_coeffs(const orc_batch* b) { return b->coeffs; }
const u64* orc_batch_leaves(const orc_batch* b) { return b->tree->leaves; }
size_t orc_batch_ncols(const orc_batch* b) { return b->ncols; }
void orc_batch_lde_row(const orc_batch* b, size_t index, size_t step, u64* out) {
    size_t j = bitrev(index * step, b->log_n + b->rate_bits);
    orc_merkle_leaf(b->tree, j, out);
}
void orc_batch_eval_ext(const orc_batch* b, const u64 zeta[2], u64* out) {
    size_t n = (size_t)1 << b->log_n;
    ext2 z = ext_make(zeta[0], zeta[1]);
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < b->ncols; ++c) {
        const u64* p = b->coeffs + c * n;
        ext2 acc = ext_from_base(0);
        for (size_t i = n; i-- > 0;) acc = ext_add(ext_mul(acc, z), ext_from_base(p[i])); /* Horner */
        out[2 * c] = acc.c[0]; out[2 * c + 1] = acc.c[1];
    }
}
void orc_batch_open(const orc_batch* b, size_t leaf_index, u64* leaf_out, u64* siblings_out) {
    orc_merkle_leaf(b->tree, leaf_index, leaf_out);
    orc_merkle_prove(b->tree, leaf_index, siblings_out);
}
orc_merkle* orc_batch_tree_(const orc_batch* b) { return b->tree; }
unsigned orc_batch_log_n_(const orc_batch* b) { return b->log_n; }

/* ---------------- Challenger ---------------- */
void orc_challenger_init(orc_challenger* c) { memset(c, 0, sizeof *c); }
static void duplexing(orc_challenger* c) {
    memcpy(c->sponge, c->input, sizeof(u64) * c->input_len); /* overwrite mode */
    c->input_len = 0;
    orc_poseidon(c->sponge);
    memcpy(c->output, c->sponge, sizeof(u64) * 8);
    c->output_len = 8;
}
void orc_challenger_observe(orc_challenger* c, const u64* e, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        c->output_len = 0; /* any buffered outputs are now invalid */
        c->input[c->input_len++] = e[i];
        if (c->input_len == 8) duplexing(c);
    }
}
u64 orc_challenger_get(orc_challenger* c) {
    if (c->input_len != 0 || c->output_len == 0) duplexing(c);
    return c->output[--c->output_len]; /* Vec::pop: state[7] first */
}
void orc_challenger_get_n(orc_challenger* c, u64* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_challenger_get(c);
}
