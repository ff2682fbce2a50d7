/* ORACLE (test infrastructure).  FRI opening proof: prover and verifier.
 * Restates plonky2 0.2.0 fri/oracle.rs (PolynomialBatch::prove_openings), util/reducing.rs (ReducingFactor),
 * fri/prover.rs (fri_proof, fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds),
 * fri/reduction_strategies.rs (ConstantArityBits), fri/verifier.rs (verify_fri_proof, fri_combine_initial,
 * compute_evaluation) -- SURVEY.md 8a rows a10/a11, Appendix A.6/A.7.  The reference reaches this code through
 * prove() at /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364 and cd.verify() at :446.
 * parity unpinned against real plonky2 output; prover and verifier are checked against each other. */
#include "vpbs_oracle.h"
#include <stdlib.h>
#include <string.h>

orc_merkle* orc_batch_tree_(const orc_batch* b);
unsigned orc_batch_log_n_(const orc_batch* b);

void orc_fri_params_standard(unsigned degree_bits, orc_fri_params* p) {
    memset(p, 0, sizeof *p);
    p->rate_bits = 3; p->cap_height = 4; p->pow_bits = 16; p->num_query_rounds = 28; p->mul_final_by_x = 0;
    /* ConstantArityBits(4, 5) */
    unsigned d = degree_bits;
    while (d > 5 && d + p->rate_bits - 4 >= p->cap_height) { p->arity_bits[p->n_rounds++] = 4; d -= 4; }
}

static unsigned final_poly_bits(const orc_fri_params* p, unsigned degree_bits) {
    unsigned d = degree_bits;
    for (unsigned i = 0; i < p->n_rounds; ++i) d -= p->arity_bits[i];
    return d;
}

size_t orc_fri_proof_words(const orc_fri_params* p, unsigned degree_bits, const size_t* ncols, size_t n_oracles) {
    unsigned log_lde = degree_bits + p->rate_bits;
    size_t cap = (size_t)4 << p->cap_height;
    size_t w = p->n_rounds * cap;
    size_t per_q = 0;
    for (size_t o = 0; o < n_oracles; ++o) per_q += ncols[o] + 4 * (size_t)(log_lde - p->cap_height);
    unsigned lg = log_lde;
    for (unsigned i = 0; i < p->n_rounds; ++i) {
        lg -= p->arity_bits[i];
        per_q += ((size_t)2 << p->arity_bits[i]) + 4 * (size_t)(lg - p->cap_height);
    }
    w += per_q * p->num_query_rounds;
    w += (size_t)2 << final_poly_bits(p, degree_bits);
    return w + 1;
}

static ext2 ch_get_ext(orc_challenger* ch) {
    u64 a = orc_challenger_get(ch), b = orc_challenger_get(ch);
    return ext_make(a, b);
}

/* componentwise coset FFT of an extension-field polynomial (twiddles and shift are base-field elements) */
static void ext_coset_fft(const ext2* coeffs, unsigned log_n, u64 shift, ext2* values) {
    size_t n = (size_t)1 << log_n;
    u64* c0 = (u64*)malloc(sizeof(u64) * n), * c1 = (u64*)malloc(sizeof(u64) * n);
    u64* v = (u64*)malloc(sizeof(u64) * n);
    for (size_t i = 0; i < n; ++i) { c0[i] = coeffs[i].c[0]; c1[i] = coeffs[i].c[1]; }
    orc_coset_lde(c0, log_n, 0, shift, v);
    for (size_t i = 0; i < n; ++i) values[i].c[0] = v[i];
    orc_coset_lde(c1, log_n, 0, shift, v);
    for (size_t i = 0; i < n; ++i) values[i].c[1] = v[i];
    free(c0); free(c1); free(v);
}

static int pow_ok(const orc_challenger* ch, u64 w, unsigned pow_bits) {
    orc_challenger c = *ch;
    orc_challenger_observe(&c, &w, 1);
    u64 r = orc_challenger_get(&c);
    return pow_bits == 0 || (r >> (64 - pow_bits)) == 0; /* leading_zeros >= pow_bits */
}

int orc_prove_openings(const orc_batch* const* oracles, size_t n_oracles, const orc_fri_batch_info* batches,
                       size_t n_batches, orc_challenger* ch, const orc_fri_params* params, unsigned degree_bits,
                       u64 forced_pow, u64* proof_out) {
    size_t n = (size_t)1 << degree_bits;
    unsigned log_lde = degree_bits + params->rate_bits;
    size_t lde = (size_t)1 << log_lde;
    for (size_t o = 0; o < n_oracles; ++o) if (orc_batch_log_n_(oracles[o]) != degree_bits) return -1;

    /* --- prove_openings: alpha-combination and quotients --- */
    ext2 alpha = ch_get_ext(ch);
    ext2* final_poly = (ext2*)calloc(lde, sizeof(ext2)); /* already zero-padded to the LDE size */
    ext2* F = (ext2*)malloc(sizeof(ext2) * n);
    for (size_t b = 0; b < n_batches; ++b) {
        const orc_fri_batch_info* bi = &batches[b];
        ext2 z = ext_make(bi->point[0], bi->point[1]);
        memset(F, 0, sizeof(ext2) * n);
        ext2 apow = ext_from_base(1); /* reduce_polys_base: powers restart at alpha^0 for every batch */
        ext2* apows = (ext2*)malloc(sizeof(ext2) * (bi->n_polys + 1));
        for (size_t j = 0; j <= bi->n_polys; ++j) { apows[j] = apow; apow = ext_mul(apow, alpha); }
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) { /* same sums, evaluated coefficient-wise so the loop parallelises */
            ext2 acc = ext_from_base(0);
            for (size_t j = 0; j < bi->n_polys; ++j) {
                const u64* c = orc_batch_coeffs(oracles[bi->oracle_index[j]]) + (size_t)bi->poly_index[j] * n;
                acc = ext_add(acc, ext_scalar_mul(apows[j], c[i]));
            }
            F[i] = acc;
        }
        apow = apows[bi->n_polys];
        free(apows);
        /* apow == alpha^n_polys == the factor applied by shift_poly */
        for (size_t i = 0; i < n; ++i) final_poly[i] = ext_mul(final_poly[i], apow);
        /* divide_by_linear(z): synthetic division, remainder dropped, top coefficient padded with zero */
        ext2 acc = ext_from_base(0);
        for (size_t i = n; i-- > 1;) {
            acc = ext_add(ext_mul(acc, z), F[i]);
            final_poly[i - 1] = ext_add(final_poly[i - 1], acc);
        }
    }
    free(F);
    if (params->mul_final_by_x) {
        for (size_t i = n; i-- > 1;) final_poly[i] = final_poly[i - 1];
        final_poly[0] = ext_from_base(0);
    }

    /* --- fri_committed_trees --- */
    ext2* coeffs = final_poly;
    size_t m = lde; unsigned log_m = log_lde;
    ext2* values = (ext2*)malloc(sizeof(ext2) * m);
    ext_coset_fft(coeffs, log_m, GL_GENERATOR, values);
    u64 shift = GL_GENERATOR;
    orc_merkle** trees = (orc_merkle**)calloc(params->n_rounds ? params->n_rounds : 1, sizeof(orc_merkle*));
    u64* w = proof_out;
    size_t cap_words = (size_t)4 << params->cap_height;
    for (unsigned r = 0; r < params->n_rounds; ++r) {
        unsigned ab = params->arity_bits[r]; size_t arity = (size_t)1 << ab;
        u64* leaves = (u64*)malloc(sizeof(u64) * 2 * m);
        for (size_t j = 0; j < m; ++j) { /* reverse_index_bits then chunk by arity, flatten */
            ext2 v = values[bitrev(j, log_m)];
            leaves[2 * j] = v.c[0]; leaves[2 * j + 1] = v.c[1];
        }
        trees[r] = orc_merkle_new(leaves, m >> ab, 2 * arity, params->cap_height);
        free(leaves);
        if (!trees[r]) return -2;
        orc_merkle_cap(trees[r], w);
        orc_challenger_observe(ch, w, cap_words);
        w += cap_words;
        ext2 beta = ch_get_ext(ch);
        size_t m2 = m >> ab;
        for (size_t i = 0; i < m2; ++i) { /* reduce_with_powers(chunk, beta) */
            ext2 acc = ext_from_base(0);
            for (size_t j = arity; j-- > 0;) acc = ext_add(ext_mul(acc, beta), coeffs[i * arity + j]);
            coeffs[i] = acc;
        }
        m = m2; log_m -= ab;
        shift = gl_exp(shift, arity);
        ext_coset_fft(coeffs, log_m, shift, values);
    }
    size_t final_len = m >> params->rate_bits;
    u64* final_words = (u64*)malloc(sizeof(u64) * 2 * final_len);
    for (size_t i = 0; i < final_len; ++i) { final_words[2 * i] = coeffs[i].c[0]; final_words[2 * i + 1] = coeffs[i].c[1]; }
    orc_challenger_observe(ch, final_words, 2 * final_len);

    /* --- fri_proof_of_work --- */
    u64 pow_witness;
    if (forced_pow != UINT64_MAX) {
        if (!pow_ok(ch, forced_pow, params->pow_bits)) return -3;
        pow_witness = forced_pow;
    } else {
        pow_witness = 0;
        if (orc_poseidon_x8_available()) {
            /* whichever of observe / get performs it, ONE duplexing happens after the witness is observed: the buffered inputs and the
             * witness (at position input_len) overwrite the state, the permutation runs, the response is state[7] -- scanned eight
             * candidates per permutation, in order, so the witness is still the smallest valid one */
            u64 pre[12];
            memcpy(pre, ch->sponge, sizeof pre);
            memcpy(pre, ch->input, sizeof(u64) * ch->input_len);
            pow_witness = orc_pow_search_x8(pre, ch->input_len, params->pow_bits, 0);
            if (!pow_ok(ch, pow_witness, params->pow_bits)) return -4;
        } else {
            while (!pow_ok(ch, pow_witness, params->pow_bits)) ++pow_witness;
        }
    }
    orc_challenger_observe(ch, &pow_witness, 1);
    (void)orc_challenger_get(ch); /* pow_response */

    /* --- fri_prover_query_rounds --- */
    for (unsigned q = 0; q < params->num_query_rounds; ++q) {
        u64 x = orc_challenger_get(ch);
        size_t x_index = (size_t)(x % lde);
        for (size_t o = 0; o < n_oracles; ++o) {
            size_t nc = orc_batch_ncols(oracles[o]);
            orc_batch_open(oracles[o], x_index, w, w + nc);
            w += nc + 4 * (size_t)(log_lde - params->cap_height);
        }
        for (unsigned r = 0; r < params->n_rounds; ++r) {
            unsigned ab = params->arity_bits[r];
            size_t leaf_len = (size_t)2 << ab;
            x_index >>= ab;
            orc_merkle_leaf(trees[r], x_index, w);
            orc_merkle_prove(trees[r], x_index, w + leaf_len);
            w += leaf_len + 4 * orc_merkle_proof_len(trees[r]);
        }
    }
    memcpy(w, final_words, sizeof(u64) * 2 * final_len); w += 2 * final_len;
    *w++ = pow_witness;
    for (unsigned r = 0; r < params->n_rounds; ++r) orc_merkle_free(trees[r]);
    free(trees); free(final_words); free(values); free(final_poly);
    return 0;
}

/* interpolate {(xs[i], ys[i])} and evaluate at t (plain Lagrange; arity <= 16) */
static ext2 interpolate_at(const ext2* xs, const ext2* ys, size_t k, ext2 t) {
    ext2 res = ext_from_base(0);
    for (size_t i = 0; i < k; ++i) {
        ext2 num = ys[i], den = ext_from_base(1);
        for (size_t j = 0; j < k; ++j) if (j != i) {
            num = ext_mul(num, ext_sub(t, xs[j]));
            den = ext_mul(den, ext_sub(xs[i], xs[j]));
        }
        res = ext_add(res, ext_mul(num, ext_inv(den)));
    }
    return res;
}

int orc_verify_fri(const u64* const* caps, const size_t* ncols, size_t n_oracles, const orc_fri_batch_info* batches,
                   const u64* const* openings, size_t n_batches, orc_challenger* ch, const orc_fri_params* params,
                   unsigned degree_bits, const u64* proof) {
    unsigned log_lde = degree_bits + params->rate_bits;
    size_t lde = (size_t)1 << log_lde;
    size_t cap_words = (size_t)4 << params->cap_height;
    size_t final_len = (size_t)1 << final_poly_bits(params, degree_bits);
    size_t total = orc_fri_proof_words(params, degree_bits, ncols, n_oracles);
    const u64* final_words = proof + total - 1 - 2 * final_len;
    u64 pow_witness = proof[total - 1];

    /* challenges (plonk/get_challenges.rs: get_fri_challenges) */
    ext2 alpha = ch_get_ext(ch);
    ext2 betas[16];
    const u64* w = proof;
    const u64* fri_caps[16];
    for (unsigned r = 0; r < params->n_rounds; ++r) {
        fri_caps[r] = w;
        orc_challenger_observe(ch, w, cap_words);
        w += cap_words;
        betas[r] = ch_get_ext(ch);
    }
    orc_challenger_observe(ch, final_words, 2 * final_len);
    orc_challenger_observe(ch, &pow_witness, 1);
    u64 pow_response = orc_challenger_get(ch);
    if (params->pow_bits && (pow_response >> (64 - params->pow_bits)) != 0) return 0;

    /* PrecomputedReducedOpenings::from_os_and_alpha */
    ext2 reduced[8];
    for (size_t b = 0; b < n_batches; ++b) {
        ext2 acc = ext_from_base(0);
        for (size_t j = batches[b].n_polys; j-- > 0;)
            acc = ext_add(ext_mul(acc, alpha), ext_make(openings[b][2 * j], openings[b][2 * j + 1]));
        reduced[b] = acc;
    }

    for (unsigned q = 0; q < params->num_query_rounds; ++q) {
        u64 x = orc_challenger_get(ch);
        size_t x_index = (size_t)(x % lde);
        /* fri_verify_initial_proof */
        const u64* leaf[8];
        for (size_t o = 0; o < n_oracles; ++o) {
            size_t nsib = log_lde - params->cap_height;
            leaf[o] = w;
            if (!orc_merkle_verify(w, ncols[o], x_index, caps[o], params->cap_height, w + ncols[o], nsib)) return 0;
            w += ncols[o] + 4 * nsib;
        }
        u64 subgroup_x = gl_mul(GL_GENERATOR, gl_exp(gl_root_of_unity(log_lde), bitrev(x_index, log_lde)));
        /* fri_combine_initial */
        ext2 sum = ext_from_base(0);
        for (size_t b = 0; b < n_batches; ++b) {
            ext2 acc = ext_from_base(0), apow = ext_from_base(1);
            for (size_t j = 0; j < batches[b].n_polys; ++j) {
                u64 e = leaf[batches[b].oracle_index[j]][batches[b].poly_index[j]];
                acc = ext_add(acc, ext_scalar_mul(apow, e));
                apow = ext_mul(apow, alpha);
            }
            ext2 num = ext_sub(acc, reduced[b]);
            ext2 den = ext_sub(ext_from_base(subgroup_x), ext_make(batches[b].point[0], batches[b].point[1]));
            sum = ext_add(ext_mul(sum, apow), ext_mul(num, ext_inv(den)));
        }
        if (params->mul_final_by_x) sum = ext_scalar_mul(sum, subgroup_x);
        ext2 old_eval = sum;
        unsigned lg = log_lde;
        for (unsigned r = 0; r < params->n_rounds; ++r) {
            unsigned ab = params->arity_bits[r]; size_t arity = (size_t)1 << ab;
            const u64* evals = w;
            size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (evals[2 * within] != old_eval.c[0] || evals[2 * within + 1] != old_eval.c[1]) return 0;
            /* compute_evaluation */
            u64 g = gl_root_of_unity(ab);
            size_t rev_within = bitrev(within, ab);
            u64 coset_start = gl_mul(subgroup_x, gl_exp(g, arity - rev_within));
            ext2 xs[16], ys[16];
            u64 y = 1;
            for (size_t i = 0; i < arity; ++i) {
                size_t src = bitrev(i, ab);
                xs[i] = ext_from_base(gl_mul(coset_start, y));
                ys[i] = ext_make(evals[2 * src], evals[2 * src + 1]);
                y = gl_mul(y, g);
            }
            old_eval = interpolate_at(xs, ys, arity, betas[r]);
            lg -= ab;
            size_t nsib = lg - params->cap_height;
            if (!orc_merkle_verify(evals, 2 * arity, coset_index, fri_caps[r], params->cap_height, evals + 2 * arity, nsib)) return 0;
            w += 2 * arity + 4 * nsib;
            for (unsigned k = 0; k < ab; ++k) subgroup_x = gl_sqr(subgroup_x);
            x_index = coset_index;
        }
        /* final_poly.eval(subgroup_x) == old_eval */
        ext2 acc = ext_from_base(0);
        for (size_t i = final_len; i-- > 0;)
            acc = ext_add(ext_scalar_mul(acc, subgroup_x), ext_make(final_words[2 * i], final_words[2 * i + 1]));
        if (!ext_eq(acc, old_eval)) return 0;
    }
    return 1;
}
