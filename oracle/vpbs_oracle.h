/* ORACLE -- CPU restatement of the vPBS proving hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (verifiable-fhe-paper_amd/) never links, imports or calls it.
 *
 * What it restates: the stages of plonky2 0.2.0 `plonk::prover::prove` that the reference reaches at
 * /root/reference/src/vtfhe/ivc_based_vpbs.rs:302-308, :333-339, :364-370 (field, FFT/LDE, Poseidon,
 * Merkle, PolynomialBatch, Challenger, openings, FRI prover AND verifier, permutation partial products, quotient polynomials with
 * the constraints of 14 gate types),
 * plus the reference's own native negacyclic NTT (/root/reference/src/vtfhe/crypto/poly.rs:9-64).
 * plonky2 0.2.0 / plonky2_field 0.2.0 / plonky2_util 0.2.0 are un-vendored crates.io dependencies
 * (/root/reference/Cargo.lock:371-374, :396-399, :421-424); their algorithm is restated from the published
 * crate.
 *
 * PARITY STATUS: Poseidon permutation pinned by the three upstream known-answer vectors
 * (tests/golden/poseidon_kat.json); negacyclic NTT pinned by the reference's TESTG/TESTGHAT vectors
 * (tests/golden/ntt_params_*.json).  Merkle/Challenger/FRI/serialisation conventions: **parity unpinned**
 * (no golden proof exists in the reference; no Rust toolchain here) -- checked for self-consistency only by the
 * verifier restated in fri.c.
 */
#ifndef VPBS_ORACLE_H
#define VPBS_ORACLE_H
#include "gl.h"

#ifdef __cplusplus
extern "C" {
#endif

void orc_set_num_threads(int n); /* OpenMP threads used by every parallel loop of the oracle */

/* ---- field helpers exported for ctypes ---- */
u64 orc_gl_add(u64 a, u64 b);
u64 orc_gl_sub(u64 a, u64 b);
u64 orc_gl_mul(u64 a, u64 b);
u64 orc_gl_inv(u64 a);
u64 orc_gl_exp(u64 a, u64 e);
u64 orc_gl_root_of_unity(unsigned k);
void orc_ext_mul(const u64 a[2], const u64 b[2], u64 out[2]);
void orc_ext_inv(const u64 a[2], u64 out[2]);

/* ---- Poseidon (hash/poseidon.rs, hash/poseidon_goldilocks.rs, hash/hashing.rs) ---- */
void orc_poseidon(u64 state[12]);
void orc_poseidon_batch(u64* states, size_t n);                 /* n independent states, [n][12] */
void orc_hash_no_pad(const u64* in, size_t n, u64 out[4]);      /* PoseidonHash::hash_no_pad   */
void orc_hash_or_noop(const u64* in, size_t n, u64 out[4]);     /* H::hash_or_noop             */
void orc_two_to_one(const u64 l[4], const u64 r[4], u64 out[4]);/* H::two_to_one               */
/* The same permutation eight at a time on AVX-512 lanes (poseidon_x8.c; naive round structure, lazy reductions): used by orc_poseidon_batch,
 * the Merkle trees and the proof-of-work scan when the CPU has AVX-512F/DQ (ORC_POSEIDON_X8=0 or orc_poseidon_x8_enable(0): scalar only).
 * Checked against orc_poseidon by tests/test_oracle_cpu.py. */
int orc_poseidon_x8_available(void);
int orc_poseidon_x8_enable(int on);
void orc_poseidon_batch_x8(u64* states, size_t n);
void orc_hash_rows_x8(const u64* rows, size_t stride, size_t len, size_t count, u64* out /* [count][4] */);
void orc_two_to_one_x8(const u64* children /* [2 count][4] */, size_t count, u64* parents /* [count][4] */);
u64 orc_pow_search_x8(const u64 state[12], unsigned pos, unsigned pow_bits, u64 start);
/* Hasher::hash_pad (plonk/config.rs): pad10*1 -- push 1, zeros until len + 1 is a multiple of the rate 8, push 1 -- then hash_no_pad */
void orc_hash_pad(const u64* in, size_t n, u64 out[4]);
/* hash chain of verify_hash_output, /root/reference/src/vtfhe/ivc_based_vpbs.rs:64-78 */
void orc_hash_chain(const u64* data, size_t n_items, size_t item_len, u64 out[4]);

/* ---- compatibility switch table: the oracle's copy of include/vpbs_prover.h `vpbs_compat` (same fields, same order, same defaults).
 *      Every restated choice of plonky2 0.2.0 that changes proof words or bytes and that no vector in this repository pins:
 *        fri_mul_final_by_x       fri/oracle.rs prove_openings / fri/verifier.rs fri_combine_initial   (0 = 0.2.0 as restated)
 *        bytes_pi_len_prefix      util/serialization write_proof_with_public_inputs                    (1)
 *        digest_domain_separator  plonk/circuit_builder.rs build(): cap || hash_pad([]) || degree_bits (1)
 *        pow_smallest_nonce       fri/prover.rs fri_proof_of_work: smallest valid nonce (forced_pow reproduces a captured one) (1)
 *      tests/step_oracle.py applies it to the transcript, the FRI parameters and the byte layout. ---- */
typedef struct {
    int fri_mul_final_by_x, bytes_pi_len_prefix, digest_domain_separator, pow_smallest_nonce;
} orc_compat;
void orc_compat_default(orc_compat* out);
/* CircuitBuilder::build's circuit_digest from the constants/sigmas cap and the degree; compat NULL = default */
void orc_circuit_digest(const orc_compat* compat, const u64* cap, size_t cap_words, unsigned degree_bits, u64 out[4]);

/* ---- FFT for proving (plonky2_field fft.rs / polynomial/mod.rs) ---- */
void orc_fft(u64* a, unsigned log_n);   /* coeffs -> values on <w_n>, natural order in and out */
void orc_ifft(u64* a, unsigned log_n);  /* values -> coeffs */
/* PolynomialCoeffs::lde(rate_bits).coset_fft(shift): out[t] = sum_i c_i (shift * w_{n<<rate}^t)^i, natural t */
void orc_coset_lde(const u64* coeffs, unsigned log_n, unsigned rate_bits, u64 shift, u64* out);

/* ---- Merkle tree with cap (hash/merkle_tree.rs, hash/merkle_proofs.rs) ---- */
typedef struct orc_merkle orc_merkle;
orc_merkle* orc_merkle_new(const u64* leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height);
void orc_merkle_free(orc_merkle*);
void orc_merkle_cap(const orc_merkle*, u64* cap_out /* [2^cap_height][4] */);
size_t orc_merkle_proof_len(const orc_merkle*);                  /* number of siblings */
void orc_merkle_leaf(const orc_merkle*, size_t idx, u64* leaf_out);
void orc_merkle_prove(const orc_merkle*, size_t idx, u64* siblings_out /* [proof_len][4] */);
int orc_merkle_verify(const u64* leaf, size_t leaf_len, size_t idx, const u64* cap, unsigned cap_height,
                      const u64* siblings, size_t n_siblings);   /* 1 = ok */

/* ---- PolynomialBatch (fri/oracle.rs) ---- */
typedef struct orc_batch orc_batch;
/* values/coeffs: column-major [ncols][1<<log_n] */
orc_batch* orc_batch_from_values(const u64* values, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height);
orc_batch* orc_batch_from_coeffs(const u64* coeffs, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height);
void orc_batch_free(orc_batch*);
void orc_batch_cap(const orc_batch*, u64* cap_out);
const u64* orc_batch_coeffs(const orc_batch*);                    /* [ncols][n] */
const u64* orc_batch_leaves(const orc_batch*);                    /* [n<<rate][ncols], plonky2 leaf order */
size_t orc_batch_ncols(const orc_batch*);
/* get_lde_values(index, step): row index*step of the natural-order LDE */
void orc_batch_lde_row(const orc_batch*, size_t index, size_t step, u64* out /* [ncols] */);
/* p.to_extension().eval(zeta) for every polynomial: out [ncols][2] */
void orc_batch_eval_ext(const orc_batch*, const u64 zeta[2], u64* out);
void orc_batch_open(const orc_batch*, size_t leaf_index, u64* leaf_out, u64* siblings_out);

/* ---- Challenger (iop/challenger.rs) ---- */
typedef struct {
    u64 sponge[12];
    u64 input[8];
    u64 output[8];
    uint32_t input_len;
    uint32_t output_len;
} orc_challenger;
void orc_challenger_init(orc_challenger*);
void orc_challenger_observe(orc_challenger*, const u64* elems, size_t n);
u64 orc_challenger_get(orc_challenger*);
void orc_challenger_get_n(orc_challenger*, u64* out, size_t n);

/* ---- FRI (fri/oracle.rs prove_openings, fri/prover.rs, fri/verifier.rs) ---- */
typedef struct {
    unsigned rate_bits;          /* 3 */
    unsigned cap_height;         /* 4 */
    unsigned pow_bits;           /* 16 */
    unsigned num_query_rounds;   /* 28 */
    unsigned n_rounds;           /* len(reduction_arity_bits) */
    unsigned arity_bits[16];
    int      mul_final_by_x;     /* 0 for plonky2 0.2.0 as recalled; switch kept until a golden proof pins it */
} orc_fri_params;
/* FriConfig::fri_params / ConstantArityBits(4,5) reduction strategy */
void orc_fri_params_standard(unsigned degree_bits, orc_fri_params* out);

typedef struct {               /* one FriBatchInfo: opening point + (oracle_index, poly_index) list */
    u64 point[2];
    size_t n_polys;
    const uint32_t* oracle_index;
    const uint32_t* poly_index;
} orc_fri_batch_info;

/* Flat FriProof layout (u64 words), identical to include/vpbs_prover.h:
 *   caps[n_rounds][2^cap_height][4]
 *   per query round q: per oracle o: leaf[ncols_o], siblings[log_lde - cap_height][4];
 *                      per fold round i: evals[2 << arity_bits_i], siblings[..][4]
 *   final_poly[len][2], pow_witness                                                         */
size_t orc_fri_proof_words(const orc_fri_params*, unsigned degree_bits, const size_t* ncols, size_t n_oracles);

/* PolynomialBatch::prove_openings -> fri_proof.  `forced_pow` : if != UINT64_MAX use this nonce
 * (the reference's rayon find_any may return any valid nonce) else take the smallest valid one. */
int orc_prove_openings(const orc_batch* const* oracles, size_t n_oracles, const orc_fri_batch_info* batches,
                       size_t n_batches, orc_challenger* ch, const orc_fri_params* params, unsigned degree_bits,
                       u64 forced_pow, u64* proof_out);
/* verify_fri_proof restated; the challenger must be in the state it had before prove_openings;
 * openings: per batch, the claimed evaluations [n_polys][2].  returns 1 when the proof verifies. */
int orc_verify_fri(const u64* const* caps, const size_t* ncols, size_t n_oracles, const orc_fri_batch_info* batches,
                   const u64* const* openings, size_t n_batches, orc_challenger* ch, const orc_fri_params* params,
                   unsigned degree_bits, const u64* proof);

/* ---- permutation argument: Z and partial products (plonk/prover.rs all_wires_permutation_partial_products,
 *      wires_permutation_partial_products_and_zs; plonk/permutation_argument.rs get_unique_coset_shifts) ----
 * wires: [>= n_routed][n] trace values; sigmas: [n_routed][n] sigma polynomial values on H (natural order);
 * k_is[j] = 7^j; subgroup x_i = w_n^i.  Per challenge c and row i: num_j = w + beta*k_j*x + gamma,
 * den_j = w + beta*sigma_j + gamma, quotients via batch inverse, products over chunks of `max_degree` (8) consecutive j,
 * running product seeded with Z(x); stored per row: the num_prods partial products then Z(x) (Z(w^0) = 1).
 * out: [num_challenges * (num_prods + 1)][n] in the prover's batch order: Z_0..Z_{nc-1}, then pp of challenge 0, 1, ... */
int orc_partial_products(const u64* wires, const u64* sigmas, size_t n_routed, unsigned log_n, const u64* betas,
                         const u64* gammas, size_t num_challenges, size_t max_degree, u64* out);

/* ---- quotient polynomials, permutation-argument part (plonk/prover.rs compute_quotient_polys, plonk/vanishing_poly.rs
 *      eval_vanishing_poly_base_batch, plonk/plonk_common.rs ZeroPolyOnCoset / reduce_with_powers_multi,
 *      util/partial_products.rs check_partial_products) ----
 * Evaluated on the coset 7<w_{8n}> (quotient_degree_factor = 8 = 2^rate_bits).  Vanishing terms, in order:
 *   L_0(x) (Z_c(x) - 1) for every challenge c; then for every challenge the (num_prods + 1) partial-product checks
 *   prev * prod(num) - next * prod(den) with prev = [Z(x), pp_0..], next = [pp_0.., Z(g x)]; then the gate-constraint
 *   terms (supplied already alpha-folded per challenge as gate_terms[c][t], natural coset order, may be NULL = none).
 * Per challenge a: q_a(x) = (sum_i term_i alpha_a^i + alpha_a^(n_terms) * gate_terms_a(x)) / Z_H(x); coset iFFT;
 * split into 8 chunks of n coefficients.  out: [num_challenges * 8][n].
 * wires/sigmas/zs_pp are COEFFICIENT matrices ([..][n]); zs_pp in batch order (Z's first). */
int orc_quotient_permutation(const u64* wires_coeffs, const u64* sigmas_coeffs, const u64* zs_pp_coeffs, size_t n_routed,
                             unsigned log_n, const u64* betas, const u64* gammas, const u64* alphas, size_t num_challenges,
                             size_t max_degree, const u64* gate_terms, u64* out);
/* verifier side of the same identity at an extension point zeta (plonk/verifier.rs + eval_vanishing_poly): given the
 * openings of the routed wires, sigmas, Z, Z(g zeta), partial products and the quotient chunks, checks for every
 * challenge  vanishing(zeta) == Z_H(zeta) * sum_m chunk_m(zeta) zeta^(n m).  gate_terms_zeta: [num_challenges][2] or NULL.
 * returns 1 when the identity holds for all challenges. */
int orc_check_vanishing_at_zeta(const u64* wires_z, const u64* sigmas_z, const u64* zs_z, const u64* zs_next_z,
                                const u64* pps_z, const u64* quotient_z, size_t n_routed, unsigned log_n,
                                const u64* betas, const u64* gammas, const u64* alphas, size_t num_challenges,
                                size_t max_degree, const u64 zeta[2], const u64* gate_terms_zeta);

/* ---- gate constraints (gates/, gates/gate.rs compute_filter, plonk/vanishing_poly.rs evaluate_gate_constraints) ----
 * kind / parameters as in gates.c; selector_index, group [start, end) and index come from the caller's restatement of
 * gates/selectors.rs selector_polynomials (tests/gates_oracle.py). */
enum { ORC_GATE_NOOP = 0, ORC_GATE_CONSTANT, ORC_GATE_PUBLIC_INPUT, ORC_GATE_ARITHMETIC, ORC_GATE_BASE_SUM, ORC_GATE_POSEIDON,
       ORC_GATE_POSEIDON_MDS, ORC_GATE_ARITHMETIC_EXT, ORC_GATE_MUL_EXT, ORC_GATE_REDUCING, ORC_GATE_REDUCING_EXT,
       ORC_GATE_RANDOM_ACCESS, ORC_GATE_EXPONENTIATION, ORC_GATE_COSET_INTERPOLATION };
typedef struct {
    unsigned kind, p0, p1, p2;
    unsigned selector_index, group_start, group_end, index;
} orc_gate;
size_t orc_gate_eval(const orc_gate* g, const ext2* wires, const ext2* gate_constants, const u64 pi_hash[4], ext2* out);
void orc_gate_terms_point(const orc_gate* gates, size_t n_gates, size_t num_selectors, const ext2* constants, const ext2* wires,
                          const u64 pi_hash[4], const u64* alphas, size_t nc, ext2* out);
/* prover side: folded gate terms on the coset 7<w_8n> in natural order, out [nc][8n]; inputs are coefficient matrices */
int orc_gate_terms_coset(const orc_gate* gates, size_t n_gates, size_t num_selectors, const u64* constants_coeffs,
                         size_t n_constants, const u64* wires_coeffs, size_t n_wires, unsigned log_n, const u64 pi_hash[4],
                         const u64* alphas, size_t nc, u64* out);
/* verifier side: from openings [..][2]; out [nc][2] */
int orc_gate_terms_zeta(const orc_gate* gates, size_t n_gates, size_t num_selectors, const u64* constants_z, size_t n_constants,
                        const u64* wires_z, size_t n_wires, const u64 pi_hash[4], const u64* alphas, size_t nc, u64* out);
/* one gate on one trace row of base-field values: out[i] == 0 iff constraint i holds; returns the constraint count */
size_t orc_gate_eval_row(const orc_gate* g, const u64* row, size_t n_wires, const u64* constants, size_t n_constants,
                         const u64 pi_hash[4], u64* out);

/* ---- negacyclic NTT of the reference (src/vtfhe/crypto/poly.rs:9-64, src/ntt/gen_param_file.sage) ---- */
void orc_negacyclic_params(unsigned log_n, u64* roots, u64* invroots, u64* ninv);
void orc_negacyclic_forward(u64* a, unsigned log_n, const u64* roots);
void orc_negacyclic_backward(u64* a, unsigned log_n, const u64* invroots, u64 ninv);

#ifdef __cplusplus
}
#endif
#endif
