/* ORACLE (test infrastructure).  Gate constraints of a plonky2 0.2.0 circuit and their alpha-folded sum.
 * Restates gates/{noop,constant,public_input,arithmetic_base,base_sum,poseidon,poseidon_mds,arithmetic_extension,
 * multiplication_extension,reducing,reducing_extension,random_access,exponentiation,coset_interpolation}.rs
 * (`eval_unfiltered`), gates/gate.rs (`eval_filtered`, `compute_filter`), plonk/vanishing_poly.rs
 * (`evaluate_gate_constraints`, the gate part of `eval_vanishing_poly[_base_batch]`) -- SURVEY.md 8a row a13; reached from
 * prove() at /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364 and cd.verify() at :446.  The gate types are those a
 * CircuitBuilder circuit under standard_recursion_config can hold (the step circuit of ivc_based_vpbs.rs:80-157).
 *
 * One code path serves the prover's base-field points and the verifier's zeta: everything is computed over GF(p^2)
 * (a base-field point is embedded as (x, 0)); wires that plonky2 reads as ExtensionAlgebra elements are pairs of GF(p^2)
 * values multiplied modulo X^2 - 7.
 * parity unpinned (wire layouts and constraint order restated from the published crate; no golden circuit here).
 * Self-consistency: every gate's generator-produced witness row must satisfy it, Poseidon rows against the pinned
 * permutation, and full proofs must verify (tests/test_oracle_cpu.py). */
#include "vpbs_oracle.h"
#include "poseidon_constants.h"
#include <stdlib.h>
#include <string.h>

typedef struct { ext2 a, b; } alg2; /* a + b X, X^2 = 7, coefficients in GF(p^2) */
static const ext2 E0 = {{0, 0}}, E1 = {{1, 0}};
static inline alg2 alg_make(ext2 a, ext2 b) { alg2 r = {a, b}; return r; }
static inline alg2 alg_add(alg2 x, alg2 y) { return alg_make(ext_add(x.a, y.a), ext_add(x.b, y.b)); }
static inline alg2 alg_sub(alg2 x, alg2 y) { return alg_make(ext_sub(x.a, y.a), ext_sub(x.b, y.b)); }
static inline alg2 alg_mul(alg2 x, alg2 y) {
    ext2 bb = ext_scalar_mul(ext_mul(x.b, y.b), 7);
    return alg_make(ext_add(ext_mul(x.a, y.a), bb), ext_add(ext_mul(x.a, y.b), ext_mul(x.b, y.a)));
}
static inline alg2 alg_scale(alg2 x, ext2 s) { return alg_make(ext_mul(x.a, s), ext_mul(x.b, s)); }
static inline alg2 alg_scale_base(alg2 x, u64 s) { return alg_make(ext_scalar_mul(x.a, s), ext_scalar_mul(x.b, s)); }
static inline alg2 walg(const ext2* w, size_t i) { return alg_make(w[i], w[i + 1]); }
static inline size_t put_alg(ext2* out, size_t k, alg2 x) { out[k] = x.a; out[k + 1] = x.b; return k + 2; }

static const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static ext2 sbox7(ext2 x) {
    ext2 x2 = ext_mul(x, x), x3 = ext_mul(x2, x), x4 = ext_mul(x2, x2);
    return ext_mul(x3, x4);
}
static void mds12(ext2* s) {
    ext2 o[12];
    for (int r = 0; r < 12; ++r) {
        ext2 acc = r == 0 ? ext_scalar_mul(s[0], 8) : E0;
        for (int i = 0; i < 12; ++i) acc = ext_add(acc, ext_scalar_mul(s[(i + r) % 12], CIRC[i]));
        o[r] = acc;
    }
    memcpy(s, o, sizeof o);
}

/* gates/poseidon.rs eval_unfiltered, with the partial rounds in the plain (not "fast") form: same S-box inputs */
static size_t poseidon_gate(const ext2* w, ext2* out) {
    size_t k = 0;
    ext2 swap = w[24], st[12];
    out[k++] = ext_mul(swap, ext_sub(swap, E1));
    for (int i = 0; i < 4; ++i) {
        ext2 delta = w[25 + i];
        out[k++] = ext_sub(ext_mul(swap, ext_sub(w[i + 4], w[i])), delta);
        st[i] = ext_add(w[i], delta);
        st[i + 4] = ext_sub(w[i + 4], delta);
    }
    for (int i = 8; i < 12; ++i) st[i] = w[i];
    for (int round = 0; round < 30; ++round) {
        for (int i = 0; i < 12; ++i) st[i] = ext_add(st[i], ext_from_base(POSEIDON_RC[12 * round + i]));
        if (round < 4 || round >= 26) {
            for (int i = 0; i < 12; ++i) {
                if (round != 0) {
                    ext2 in = round < 4 ? w[29 + 12 * (round - 1) + i] : w[87 + 12 * (round - 26) + i];
                    out[k++] = ext_sub(st[i], in);
                    st[i] = in;
                }
                st[i] = sbox7(st[i]);
            }
        } else {
            ext2 in = w[65 + (round - 4)];
            out[k++] = ext_sub(st[0], in);
            st[0] = sbox7(in);
        }
        mds12(st);
    }
    for (int i = 0; i < 12; ++i) out[k++] = ext_sub(w[12 + i], st[i]);
    return k;
}

static void subgroup_and_weights(unsigned bits, u64* dom, u64* wts) {
    size_t n = (size_t)1 << bits;
    u64 g = gl_root_of_unity(bits), x = 1;
    for (size_t i = 0; i < n; ++i) { dom[i] = x; x = gl_mul(x, g); }
    for (size_t i = 0; i < n; ++i) {
        u64 d = 1;
        for (size_t j = 0; j < n; ++j) if (j != i) d = gl_mul(d, gl_sub(dom[i], dom[j]));
        wts[i] = gl_inv(d);
    }
}

/* constraints of one gate at one point -> out[0 .. return) */
size_t orc_gate_eval(const orc_gate* g, const ext2* w, const ext2* c, const u64 pi_hash[4], ext2* out) {
    size_t k = 0;
    switch (g->kind) {
    case ORC_GATE_NOOP: break;
    case ORC_GATE_CONSTANT:
        for (unsigned i = 0; i < g->p0; ++i) out[k++] = ext_sub(c[i], w[i]);
        break;
    case ORC_GATE_PUBLIC_INPUT:
        for (unsigned i = 0; i < 4; ++i) out[k++] = ext_sub(w[i], ext_from_base(pi_hash[i]));
        break;
    case ORC_GATE_ARITHMETIC:
        for (unsigned i = 0; i < g->p0; ++i) {
            ext2 prod = ext_mul(ext_mul(w[4 * i], w[4 * i + 1]), c[0]);
            out[k++] = ext_sub(w[4 * i + 3], ext_add(prod, ext_mul(w[4 * i + 2], c[1])));
        }
        break;
    case ORC_GATE_BASE_SUM: {
        ext2 acc = E0;
        for (unsigned i = g->p0; i-- > 0;) acc = ext_add(ext_scalar_mul(acc, g->p1), w[1 + i]);
        out[k++] = ext_sub(acc, w[0]);
        for (unsigned i = 0; i < g->p0; ++i) {
            ext2 prod = E1;
            for (unsigned d = 0; d < g->p1; ++d) prod = ext_mul(prod, ext_sub(w[1 + i], ext_from_base(d)));
            out[k++] = prod;
        }
        break;
    }
    case ORC_GATE_POSEIDON: k = poseidon_gate(w, out); break;
    case ORC_GATE_POSEIDON_MDS:
        for (unsigned r = 0; r < 12; ++r) {
            alg2 acc = r == 0 ? alg_scale_base(walg(w, 0), 8) : alg_make(E0, E0);
            for (unsigned i = 0; i < 12; ++i) acc = alg_add(acc, alg_scale_base(walg(w, 2 * ((i + r) % 12)), CIRC[i]));
            k = put_alg(out, k, alg_sub(walg(w, 24 + 2 * r), acc));
        }
        break;
    case ORC_GATE_ARITHMETIC_EXT:
        for (unsigned i = 0; i < g->p0; ++i) {
            alg2 m = alg_scale(alg_mul(walg(w, 8 * i), walg(w, 8 * i + 2)), c[0]);
            alg2 computed = alg_add(m, alg_scale(walg(w, 8 * i + 4), c[1]));
            k = put_alg(out, k, alg_sub(walg(w, 8 * i + 6), computed));
        }
        break;
    case ORC_GATE_MUL_EXT:
        for (unsigned i = 0; i < g->p0; ++i)
            k = put_alg(out, k, alg_sub(walg(w, 6 * i + 4), alg_scale(alg_mul(walg(w, 6 * i), walg(w, 6 * i + 2)), c[0])));
        break;
    case ORC_GATE_REDUCING:
    case ORC_GATE_REDUCING_EXT: {
        int ext = g->kind == ORC_GATE_REDUCING_EXT;
        unsigned n = g->p0, start_accs = ext ? 6 + 2 * n : 6 + n;
        alg2 alpha = walg(w, 2), acc = walg(w, 4);
        for (unsigned i = 0; i < n; ++i) {
            alg2 coeff = ext ? walg(w, 6 + 2 * i) : alg_make(w[6 + i], E0);
            alg2 next = i == n - 1 ? walg(w, 0) : walg(w, start_accs + 2 * i);
            k = put_alg(out, k, alg_sub(alg_add(alg_mul(acc, alpha), coeff), next));
            acc = next;
        }
        break;
    }
    case ORC_GATE_RANDOM_ACCESS: {
        unsigned bits = g->p0, copies = g->p1, extra = g->p2, vec = 1u << bits;
        unsigned routed = (2 + vec) * copies + extra;
        for (unsigned cp = 0; cp < copies; ++cp) {
            const ext2* b = w + routed + cp * bits;
            const ext2* base = w + (2 + vec) * cp;
            for (unsigned i = 0; i < bits; ++i) out[k++] = ext_mul(b[i], ext_sub(b[i], E1));
            ext2 idx = E0;
            for (unsigned i = bits; i-- > 0;) idx = ext_add(ext_add(idx, idx), b[i]);
            out[k++] = ext_sub(idx, base[0]);
            ext2 items[32];
            for (unsigned i = 0; i < vec; ++i) items[i] = base[2 + i];
            for (unsigned i = 0, len = vec / 2; i < bits; ++i, len /= 2)
                for (unsigned j = 0; j < len; ++j)
                    items[j] = ext_add(items[2 * j], ext_mul(b[i], ext_sub(items[2 * j + 1], items[2 * j])));
            out[k++] = ext_sub(items[0], base[1]);
        }
        for (unsigned i = 0; i < extra; ++i) out[k++] = ext_sub(c[i], w[(2 + vec) * copies + i]);
        break;
    }
    case ORC_GATE_EXPONENTIATION: {
        unsigned n = g->p0;
        ext2 prev = E1;
        for (unsigned i = 0; i < n; ++i) {
            ext2 sq = i == 0 ? E1 : ext_mul(prev, prev);
            ext2 bit = w[1 + (n - 1 - i)];
            ext2 computed = ext_mul(sq, ext_add(ext_mul(bit, w[0]), ext_sub(E1, bit)));
            out[k++] = ext_sub(computed, w[2 + n + i]);
            prev = w[2 + n + i];
        }
        out[k++] = ext_sub(w[1 + n], prev);
        break;
    }
    case ORC_GATE_COSET_INTERPOLATION: {
        unsigned bits = g->p0, degree = g->p1, points = 1u << bits, ni = (points - 2) / (degree - 1);
        unsigned s_point = 1 + 2 * points, s_value = s_point + 2, s_inter = s_value + 2, s_shifted = s_inter + 4 * ni;
        u64 dom[32], wts[32];
        subgroup_and_weights(bits, dom, wts);
        alg2 shifted = walg(w, s_shifted);
        k = put_alg(out, k, alg_sub(walg(w, s_point), alg_scale(shifted, w[0])));
        alg2 eval = alg_make(E0, E0), prod = alg_make(E1, E0);
        unsigned from = 0, to = degree < points ? degree : points;
        for (unsigned chunk = 0;; ++chunk) {
            for (unsigned i = from; i < to; ++i) { /* partial_interpolate */
                alg2 term = shifted;
                term.a = ext_sub(term.a, ext_from_base(dom[i]));
                eval = alg_add(alg_mul(eval, term), alg_mul(alg_scale_base(walg(w, 1 + 2 * i), wts[i]), prod));
                prod = alg_mul(prod, term);
            }
            if (chunk == ni) break;
            alg2 ie = walg(w, s_inter + 2 * chunk), ip = walg(w, s_inter + 2 * (ni + chunk));
            k = put_alg(out, k, alg_sub(ie, eval));
            k = put_alg(out, k, alg_sub(ip, prod));
            eval = ie; prod = ip;
            from = 1 + (degree - 1) * (chunk + 1);
            to = from + degree - 1 < points ? from + degree - 1 : points;
        }
        k = put_alg(out, k, alg_sub(walg(w, s_value), eval));
        break;
    }
    default: break;
    }
    return k;
}

/* sum_i alpha^i sum_g filter_g c_{g,i} at one point.  constants: every constants column (selectors first). */
void orc_gate_terms_point(const orc_gate* gates, size_t n_gates, size_t num_selectors, const ext2* constants, const ext2* wires,
                          const u64 pi_hash[4], const u64* alphas, size_t nc, ext2* out) {
    ext2 total[256], tmp[256];
    size_t max_c = 0;
    for (size_t k = 0; k < 256; ++k) total[k] = E0;
    for (size_t gi = 0; gi < n_gates; ++gi) {
        const orc_gate* g = &gates[gi];
        size_t cnt = orc_gate_eval(g, wires, constants + num_selectors, pi_hash, tmp);
        /* compute_filter */
        ext2 s = constants[g->selector_index], f = E1;
        for (unsigned i = g->group_start; i < g->group_end; ++i)
            if (i != g->index) f = ext_mul(f, ext_sub(ext_from_base(i), s));
        if (num_selectors > 1) f = ext_mul(f, ext_sub(ext_from_base(0xFFFFFFFFu), s));
        for (size_t k = 0; k < cnt; ++k) total[k] = ext_add(total[k], ext_mul(f, tmp[k]));
        if (cnt > max_c) max_c = cnt;
    }
    for (size_t a = 0; a < nc; ++a) {
        ext2 acc = E0;
        for (size_t k = max_c; k-- > 0;) acc = ext_add(ext_scalar_mul(acc, alphas[a]), total[k]);
        out[a] = acc;
    }
}

/* on the coset 7<w_8n>, natural order, from coefficient matrices: out [nc][8n] */
int orc_gate_terms_coset(const orc_gate* gates, size_t n_gates, size_t num_selectors, const u64* constants_coeffs,
                         size_t n_constants, const u64* wires_coeffs, size_t n_wires, unsigned log_n, const u64 pi_hash[4],
                         const u64* alphas, size_t nc, u64* out) {
    const unsigned rate_bits = 3;
    size_t n = (size_t)1 << log_n, big = n << rate_bits;
    u64* C = (u64*)malloc(sizeof(u64) * (n_constants ? n_constants : 1) * big);
    u64* W = (u64*)malloc(sizeof(u64) * n_wires * big);
    if (!C || !W) { free(C); free(W); return -1; }
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t j = 0; j < n_constants + n_wires; ++j) {
        if (j < n_constants) orc_coset_lde(constants_coeffs + j * n, log_n, rate_bits, GL_GENERATOR, C + j * big);
        else orc_coset_lde(wires_coeffs + (j - n_constants) * n, log_n, rate_bits, GL_GENERATOR, W + (j - n_constants) * big);
    }
    int bad = 0;
#pragma omp parallel for schedule(static)
    for (size_t t = 0; t < big; ++t) {
        ext2 cw[16], ww[160], res[8];
        if (n_constants > 16 || n_wires > 160 || nc > 8) { bad = 1; continue; }
        for (size_t j = 0; j < n_constants; ++j) cw[j] = ext_from_base(C[j * big + t]);
        for (size_t j = 0; j < n_wires; ++j) ww[j] = ext_from_base(W[j * big + t]);
        for (size_t j = n_wires; j < 160; ++j) ww[j] = E0;
        orc_gate_terms_point(gates, n_gates, num_selectors, cw, ww, pi_hash, alphas, nc, res);
        for (size_t a = 0; a < nc; ++a) {
            if (res[a].c[1] != 0) bad = 1; /* base-field inputs must give base-field results */
            out[a * big + t] = res[a].c[0];
        }
    }
    free(C); free(W);
    return bad ? -2 : 0;
}

/* verifier side: from the openings at zeta ([..][2] arrays) */
int orc_gate_terms_zeta(const orc_gate* gates, size_t n_gates, size_t num_selectors, const u64* constants_z, size_t n_constants,
                        const u64* wires_z, size_t n_wires, const u64 pi_hash[4], const u64* alphas, size_t nc, u64* out) {
    ext2 cw[16], ww[160], res[8];
    if (n_constants > 16 || n_wires > 160 || nc > 8) return -1;
    for (size_t j = 0; j < n_constants; ++j) cw[j] = ext_make(constants_z[2 * j], constants_z[2 * j + 1]);
    for (size_t j = 0; j < 160; ++j) ww[j] = j < n_wires ? ext_make(wires_z[2 * j], wires_z[2 * j + 1]) : E0;
    orc_gate_terms_point(gates, n_gates, num_selectors, cw, ww, pi_hash, alphas, nc, res);
    for (size_t a = 0; a < nc; ++a) { out[2 * a] = res[a].c[0]; out[2 * a + 1] = res[a].c[1]; }
    return 0;
}

/* constraints of one gate on one trace row of base-field values (tests: generator rows must give all zeros) */
size_t orc_gate_eval_row(const orc_gate* g, const u64* row, size_t n_wires, const u64* constants, size_t n_constants,
                         const u64 pi_hash[4], u64* out) {
    ext2 cw[16], ww[160], res[256];
    for (size_t j = 0; j < 16; ++j) cw[j] = j < n_constants ? ext_from_base(constants[j]) : E0;
    for (size_t j = 0; j < 160; ++j) ww[j] = j < n_wires ? ext_from_base(row[j]) : E0;
    size_t k = orc_gate_eval(g, ww, cw, pi_hash, res);
    for (size_t i = 0; i < k; ++i) out[i] = res[i].c[0] | res[i].c[1]; /* zero iff both components are zero */
    return k;
}
