/* vpbs_prover.h -- C ABI of the MI355X-native prover for the vPBS step circuit.
 *
 * Drop-in boundary.  plonky2 0.2.0 has no plugin API; the reference reaches the hot path through
 *   plonky2::plonk::prover::prove::<F, C, D>(&prover_only, &common, pw, &mut timing)
 * at /root/reference/src/vtfhe/ivc_based_vpbs.rs:302-308, :333-339, :364-370 (and CircuitData::prove in every test,
 * e.g. /root/reference/src/ntt/mod.rs:99).  The seam a device backend replaces sits one level below, inside the
 * un-vendored crate (pinned at /root/reference/Cargo.lock:371-374): fri/oracle.rs `PolynomialBatch::from_values`,
 * `from_coeffs`, `get_lde_values`, `prove_openings` and plonk/proof.rs `OpeningSet::new`.  Each entry point below
 * names the plonky2 function it replaces.  INTEGRATION.md shows the Rust `extern "C"` binding.
 *
 * Conventions: field elements are canonical u64 (< p = 2^64 - 2^32 + 1); GF(p^2) elements are [c0, c1]; polynomial
 * matrices are column-major [col][row]; hashes are 4 u64.  Every function returns 0 on success and a negative
 * vpbs_status on error (no exceptions cross the boundary; vpbs_last_error(ctx) has the text).  Host buffers are owned
 * by the caller; vpbs_batch handles are device-resident and owned by the library until vpbs_batch_free.
 * A vpbs_ctx is bound to one device and one HIP stream and is NOT re-entrant: use one ctx per host thread.
 * `_dev` variants take device pointers valid on the ctx's device and enqueue on the ctx's stream.
 * A HIP failure pending on the calling thread (a launch refused for its configuration or for resources -- this library's or another one's)
 * is reported by the first call of this library that waits on the device or ends a stage, as VPBS_ERR_DEVICE; the call that reports it
 * also clears it from the thread's HIP error state: one call fails, the next ones start clean, and the host needs no hipGetLastError.
 */
#ifndef VPBS_PROVER_H
#define VPBS_PROVER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vpbs_ctx vpbs_ctx;
typedef struct vpbs_batch vpbs_batch;

typedef enum {
    VPBS_OK = 0,
    VPBS_ERR_INVALID = -1,   /* bad argument (size not a power of two, null pointer, ...) */
    VPBS_ERR_DEVICE = -2,    /* HIP runtime error (no device, launch failure, ...) */
    VPBS_ERR_OOM = -3,       /* device allocation failed */
    VPBS_ERR_POW = -4,       /* forced proof-of-work nonce is not valid / search exhausted */
    VPBS_ERR_PEER = -5       /* a sharded step proof: ANOTHER rank of the communicator failed (its own call returns its own error); every
                                rank of the step returns an error, none hangs in a collective */
} vpbs_status;

#define VPBS_POW_ANY UINT64_MAX

/* ---- context ---- */
int vpbs_ctx_create(int device_ordinal, unsigned log_n_max, unsigned rate_bits, unsigned cap_height, vpbs_ctx** out);
void vpbs_ctx_destroy(vpbs_ctx* ctx);
const char* vpbs_last_error(const vpbs_ctx* ctx);
int vpbs_ctx_synchronize(vpbs_ctx* ctx);
/* Streams used for the gate-constraint stage of a step proof: 1 (default) keeps the gate kernel and the permutation part of the quotient on
 * the context's stream -- right for the LDS-tile gate kernel, whose workgroups fill a CU by themselves; 3 spreads the per-gate launches and
 * the permutation part over two helper streams (the better arrangement for VPBS_OPT_GATES_FUSED = 0 / VPBS_OPT_GATES_TILE = 0). */
int vpbs_ctx_set_gate_lanes(vpbs_ctx* ctx, unsigned lanes);
void* vpbs_ctx_stream(vpbs_ctx* ctx); /* hipStream_t, for callers that share device buffers with the ctx */
/* Launch heuristics of a context.  Every option has an environment variable of the same meaning that only sets the DEFAULT a new context
 * starts with (read at vpbs_ctx_create); results never depend on any of them -- they choose between bit-identical kernel arrangements. */
typedef enum {
    VPBS_OPT_GATE_LANES = 0,      /* 1 (default) | 3: streams of the gate-constraint stage (= vpbs_ctx_set_gate_lanes).  VPBS_GATE_LANES     */
    VPBS_OPT_GATES_FUSED = 1,     /* 1 (default): all gate constraints in one launch; 0: one launch per gate type.       VPBS_GATES_FUSED    */
    VPBS_OPT_GATE_ITEMS = 2,      /* reserved: tuned the (tile x item) gate kernel of rounds 2-4, removed in round 6; 1..8 is still
                                     accepted and read back, and has no effect.                                          VPBS_GATE_ITEMS     */
    VPBS_OPT_WIDE_THRESHOLD = 3,  /* launches with at most this many independent permutations (Merkle / FRI tree levels, FRI leaves) use
                                     the 16-lane Poseidon form: a third of the latency for 3.7 x the instructions.  Default 2^14: right
                                     for a context that has the GPU to itself.  A context that SHARES the GPU with other chains should
                                     lower it (2048): the other chains hide the latency, only the instruction count is left
                                     (six chains: 8.33 -> 8.14-8.22 ms per chained proof).  Below the default the proof-of-work search
                                     also runs its range in rounds of 2^15 candidates that stop once a smaller nonce is known (a third
                                     fewer permutations, each round one permutation deep) instead of all at once.        VPBS_WIDE_THRESHOLD */
    VPBS_OPT_MERKLE_CLIMB = 4,    /* 1 (default): the upper levels of a tree in fused multi-level launches; 0: per level. VPBS_MERKLE_CLIMB  */
    VPBS_OPT_GATES_TILE = 5       /* 1 (default): the one-launch gate kernel stages a 64-point tile of every column in LDS and its eight
                                     waves share the gates; 0: one launch per gate type, as VPBS_OPT_GATES_FUSED = 0 (also what a gate
                                     set that does not fit a tile plan gets).                                            VPBS_GATES_TILE */
} vpbs_option;
int vpbs_ctx_set_option(vpbs_ctx* ctx, int option, uint64_t value);
int vpbs_ctx_get_option(const vpbs_ctx* ctx, int option, uint64_t* value_out);
/* Host side (process-wide): independent Poseidon permutations eight at a time on AVX-512 lanes (witness generation, the verifier's Merkle
 * paths) where the CPU has them.  on = 0 forces the scalar form (A/B measurements, tests); returns what is in force.  Default: on, or the
 * environment variable VPBS_POSEIDON_X8. */
int vpbs_host_set_poseidon_x8(int on);
/* the FriConfig shape the context was created for (vpbs_ctx_create's rate_bits / cap_height); 0 for a null context */
unsigned vpbs_ctx_rate_bits(const vpbs_ctx* ctx);
unsigned vpbs_ctx_cap_height(const vpbs_ctx* ctx);
int vpbs_ctx_device(const vpbs_ctx* ctx);   /* the device ordinal the context was created on; -1 for a null context */

/* ---- compatibility switch table ----
 * plonky2 0.2.0 is an un-vendored dependency of the reference (/root/reference/Cargo.lock:371-374) and cannot be run in the authoring image,
 * so a few transcript / layout choices of the crate are restated without a golden proof to pin them.  Every such choice that changes a
 * proof's words or bytes sits in THIS table, and prover, serialiser, verifier and the test oracle (oracle/vpbs_oracle.h `orc_compat`) all
 * read it: one capture of a real proof (tools/plonky2_capture) is checked against every position, and to_fixture.py records the one that
 * reproduces it.  The defaults are plonky2 0.2.0 as restated (vpbs_compat_default); the alternative of each switch is the behaviour of
 * earlier releases of the crate.  Call sites in the reference: proof.to_bytes() ivc_based_vpbs.rs:488, cd.verify :443-447,
 * add_verifier_data_public_inputs :209-214 (the circuit digest travels in the public inputs of every step proof). */
typedef struct {
    int fri_mul_final_by_x;       /* fri/oracle.rs prove_openings.  0 (default): each batch quotient is padded back to a power of two
                                     ("quotient.coeffs.push(ZERO)").  1: the final polynomial is multiplied by X before the LDE and
                                     fri_combine_initial multiplies its sum by subgroup_x (releases before the padding change) */
    int bytes_pi_len_prefix;      /* util/serialization write_proof_with_public_inputs.  1 (default): write_usize(public_inputs.len()) as a
                                     little-endian u64 in front of the public inputs.  0: the public inputs run to the end of the buffer
                                     (the older Buffer reader: remaining / 8 elements) */
    int digest_domain_separator;  /* plonk/circuit_builder.rs build().  1 (default): circuit_digest = hash_no_pad(constants_sigmas_cap ||
                                     hash_pad(domain_separator = []) || degree_bits).  0: hash_no_pad(cap || degree_bits) (before the
                                     domain separator existed; what rounds 1-2 of this library used) */
    int pow_smallest_nonce;       /* fri/prover.rs fri_proof_of_work.  1 (default): the SMALLEST u64 whose challenger response has the
                                     required leading zeros -- deterministic.  The crate's rayon find_any may return ANY valid nonce, so a
                                     captured proof is reproduced by passing its pow_witness as vpbs_step_inputs.forced_pow; 0 is reserved
                                     for "first found" and is rejected by this build */
} vpbs_compat;
void vpbs_compat_default(vpbs_compat* out);
/* A context proves and serialises under one table (default at creation); VPBS_ERR_INVALID for a position this build does not implement. */
int vpbs_ctx_set_compat(vpbs_ctx* ctx, const vpbs_compat* compat);
int vpbs_ctx_get_compat(const vpbs_ctx* ctx, vpbs_compat* out);

/* ---- PolynomialBatch (plonky2 fri/oracle.rs) ---- */
/* = PolynomialBatch::from_values(values, rate_bits, blinding=false, cap_height, ..): iFFT -> coset LDE -> Merkle */
int vpbs_commit_values(vpbs_ctx* ctx, const uint64_t* values, unsigned ncols, unsigned log_n, vpbs_batch** out,
                       uint64_t* cap_out /* [2^cap_height][4] */);
/* = PolynomialBatch::from_coeffs */
int vpbs_commit_coeffs(vpbs_ctx* ctx, const uint64_t* coeffs, unsigned ncols, unsigned log_n, vpbs_batch** out,
                       uint64_t* cap_out);
int vpbs_commit_values_dev(vpbs_ctx* ctx, const uint64_t* d_values, unsigned ncols, unsigned log_n, vpbs_batch** out,
                           uint64_t* cap_out);
int vpbs_commit_coeffs_dev(vpbs_ctx* ctx, const uint64_t* d_coeffs, unsigned ncols, unsigned log_n, vpbs_batch** out,
                           uint64_t* cap_out);
/* Multi-GPU coset sharding: rank `shard` of `n_shards` (a power of two <= 2^rate_bits) computes only
 * its cosets of the LDE -- the contiguous leaf range [shard, shard+1) * (n << rate_bits) / n_shards -- hashes those
 * leaves and builds its 2^cap_height / n_shards cap subtrees.  local_cap_out receives those cap entries; the full cap is
 * the concatenation over ranks (one all-gather of 32-byte hashes: RCCL over xGMI in bench/production, gloo in the CPU
 * tests).  The returned batch holds the shard only: vpbs_batch_open works for its own leaves; FRI over sharded
 * oracles is not part of this round.  is_values: 1 = from_values (iNTT first), 0 = from_coeffs. */
int vpbs_commit_sharded_dev(vpbs_ctx* ctx, const uint64_t* d_data, int is_values, unsigned ncols, unsigned log_n,
                            unsigned shard, unsigned n_shards, vpbs_batch** out, uint64_t* local_cap_out);
void vpbs_batch_free(vpbs_batch* batch);
unsigned vpbs_batch_ncols(const vpbs_batch* batch);
unsigned vpbs_batch_log_n(const vpbs_batch* batch);
int vpbs_batch_cap(vpbs_batch* batch, uint64_t* cap_out);
/* batch.polynomials: coefficient form, [ncols][n] */
int vpbs_batch_coeffs(vpbs_batch* batch, uint64_t* out);
/* = get_lde_values(index, step) for index = row_start .. row_start+nrows-1: out[k][c] = lde_c[(row_start+k)*step],
 * natural LDE order (for a host-side quotient evaluation) */
int vpbs_batch_lde_rows(vpbs_batch* batch, size_t row_start, size_t nrows, size_t step, uint64_t* out);
/* = the commitment's share of OpeningSet::new: out[c] = polynomials[c].to_extension().eval(zeta) */
int vpbs_batch_eval_ext(vpbs_batch* batch, const uint64_t zeta[2], uint64_t* out /* [ncols][2] */);
/* = (merkle_tree.get(leaf_index), merkle_tree.prove(leaf_index)) */
int vpbs_batch_open(vpbs_batch* batch, size_t leaf_index, uint64_t* leaf_out /* [ncols] */,
                    uint64_t* siblings_out /* [log_lde - cap_height][4] */);

/* ---- Challenger (plonky2 iop/challenger.rs); the state crosses the boundary explicitly ---- */
typedef struct {
    uint64_t sponge[12];
    uint64_t input[8];
    uint64_t output[8];
    uint32_t input_len;
    uint32_t output_len;
} vpbs_challenger_state;
void vpbs_challenger_init(vpbs_challenger_state* ch);
void vpbs_challenger_observe(vpbs_challenger_state* ch, const uint64_t* elems, size_t n); /* observe_elements */
uint64_t vpbs_challenger_get(vpbs_challenger_state* ch);                                 /* get_challenge   */
/* PoseidonHash::hash_no_pad on the host (public-input hash, small inputs) */
void vpbs_hash_no_pad(const uint64_t* in, size_t n, uint64_t out[4]);
/* PoseidonHash::hash_pad (plonk/config.rs Hasher::hash_pad): the pad10*1 rule -- push 1, zeros until len + 1 is a multiple of the sponge
 * rate 8, push 1 -- then hash_no_pad.  CircuitBuilder::build hashes the (empty) domain separator with it. */
void vpbs_hash_pad(const uint64_t* in, size_t n, uint64_t out[4]);
/* verifier_only.circuit_digest as CircuitBuilder::build derives it from the constants/sigmas cap and the degree (formula: vpbs_compat.
 * digest_domain_separator; compat NULL = default).  cap: [cap_words] = 2^cap_height hashes.  The digest is the first thing the transcript
 * absorbs and, in the IVC chain, part of every step's public inputs (ivc_based_vpbs.rs:209-214, check_cyclic_proof_verifier_data :448-452). */
int vpbs_circuit_digest(const vpbs_compat* compat, const uint64_t* cap, size_t cap_words, unsigned degree_bits, uint64_t out[4]);
/* the hash chain of verify_hash_output (/root/reference/src/vtfhe/ivc_based_vpbs.rs:64-78): h_0 = 0^4,
 * h_{k+1} = hash_no_pad(h_k || item_k) over n_items items of item_len elements each (row-major); host only.
 * returns 1 if the chain ends in `claimed` (or claimed == NULL: just computes), 0 otherwise; out may be NULL. */
int vpbs_hash_chain(const uint64_t* items, size_t n_items, size_t item_len, const uint64_t claimed[4], uint64_t out[4]);
/* Links of such a chain from any point: out[k] = hash_no_pad(h_{k-1} || items[k]), h_{-1} = prefix, k < n_links (items[k]: item_len
 * elements; out: [n_links][4]); host only.  Calls in progress at the same time with the same n_links and item_len (the hash-chain threads
 * of several vPBS chains in one process) share the lanes of the eight-lane host Poseidon -- one chain per lane, all links in lockstep --
 * where the CPU has AVX-512, the items are 64 elements or longer and the process is short of CPUs (the mode of
 * vpbs_host_set_blocking_sync: a batch runs on one thread while the other callers sleep): 0.41 instead of 1.28 us per permutation with
 * eight callers.  The result never depends on who shares. */
int vpbs_hash_chain_links(const uint64_t prefix[4], const uint64_t* const* items, size_t n_links, size_t item_len, uint64_t* out);

/* ---- FRI (plonky2 fri/oracle.rs prove_openings -> fri/prover.rs fri_proof) ---- */
typedef struct {
    unsigned rate_bits;        /* 3  */
    unsigned cap_height;       /* 4  */
    unsigned pow_bits;         /* 16 */
    unsigned num_query_rounds; /* 28 */
    unsigned n_rounds;         /* len(reduction_arity_bits) */
    unsigned arity_bits[16];
    int mul_final_by_x;        /* vpbs_compat.fri_mul_final_by_x for a direct vpbs_fri_prove call; vpbs_prove_step and the verifier take
                                  the switch from their compat table, not from here */
} vpbs_fri_params;
/* CircuitConfig::standard_recursion_config().fri_config.fri_params(degree_bits, hiding=false); mul_final_by_x = the default (0) */
void vpbs_fri_params_standard(unsigned degree_bits, vpbs_fri_params* out);

typedef struct { /* FriBatchInfo: opening point + FriPolynomialInfo list */
    uint64_t point[2];
    size_t n_polys;
    const uint32_t* oracle_index;
    const uint32_t* poly_index;
} vpbs_fri_batch_info;
typedef struct { /* FriInstanceInfo (oracles are passed separately) */
    const vpbs_fri_batch_info* batches;
    size_t n_batches;
} vpbs_fri_instance;

/* FriProof as flat u64 words, in plonky2's serialisation order (Merkle-proof length bytes omitted):
 *   commit_phase_merkle_caps[n_rounds][2^cap_height][4]
 *   query_round_proofs[q]: initial_trees_proof: per oracle { leaf[ncols_o], siblings[log_lde-cap_height][4] }
 *                          steps[i]: { evals[2 << arity_bits_i], siblings[log(len_i) - cap_height][4] }
 *   final_poly[len][2], pow_witness */
size_t vpbs_fri_proof_words(const vpbs_fri_params* params, unsigned degree_bits, const size_t* ncols, size_t n_oracles);
/* forced_pow: VPBS_POW_ANY -> smallest valid nonce; otherwise use the given nonce (the reference's rayon find_any
 * may return any valid nonce, so byte parity with a reference proof is conditional on its pow_witness). */
int vpbs_fri_prove(vpbs_ctx* ctx, vpbs_batch* const* oracles, size_t n_oracles, const vpbs_fri_instance* instance,
                   const vpbs_fri_params* params, vpbs_challenger_state* challenger /* inout */, uint64_t forced_pow,
                   uint64_t* proof_out);

/* ---- gate constraints (plonky2 0.2.0 gates/ `eval_unfiltered`, gates/gate.rs `eval_filtered_base_batch` /
 *      `compute_filter`, gates/selectors.rs `selector_polynomials`, plonk/vanishing_poly.rs
 *      `evaluate_gate_constraints_base_batch`).  The gate types are the ones a circuit built with
 *      CircuitBuilder + standard_recursion_config can contain (/root/reference/src/vtfhe/ivc_based_vpbs.rs:80-157 builds the
 *      step circuit from arithmetic, base-sum, Poseidon and the recursive-verifier gadgets).  Restated from the published
 *      crate: wire layouts, constraint order and the id strings used for sorting are parity-unpinned (no golden circuit). */
typedef enum {
    VPBS_GATE_NOOP = 0,            /* NoopGate */
    VPBS_GATE_CONSTANT,            /* ConstantGate { num_consts = p0 } */
    VPBS_GATE_PUBLIC_INPUT,        /* PublicInputGate */
    VPBS_GATE_ARITHMETIC,          /* ArithmeticGate { num_ops = p0 } */
    VPBS_GATE_BASE_SUM,            /* BaseSumGate<B = p1> { num_limbs = p0 } */
    VPBS_GATE_POSEIDON,            /* PoseidonGate */
    VPBS_GATE_POSEIDON_MDS,        /* PoseidonMdsGate */
    VPBS_GATE_ARITHMETIC_EXT,      /* ArithmeticExtensionGate { num_ops = p0 } */
    VPBS_GATE_MUL_EXT,             /* MulExtensionGate { num_ops = p0 } */
    VPBS_GATE_REDUCING,            /* ReducingGate { num_coeffs = p0 } */
    VPBS_GATE_REDUCING_EXT,        /* ReducingExtensionGate { num_coeffs = p0 } */
    VPBS_GATE_RANDOM_ACCESS,       /* RandomAccessGate { bits = p0, num_copies = p1, num_extra_constants = p2 } */
    VPBS_GATE_EXPONENTIATION,      /* ExponentiationGate { num_power_bits = p0 } */
    VPBS_GATE_COSET_INTERPOLATION, /* CosetInterpolationGate { subgroup_bits = p0, degree = p1 } */
    VPBS_GATE_KINDS
} vpbs_gate_kind;
typedef struct {
    unsigned kind, p0, p1, p2;        /* set by the caller */
    /* filled by vpbs_gates_layout: */
    unsigned degree, num_constraints, num_constants, num_wires;
    unsigned selector_index;          /* which selector polynomial (leading constants column) carries this gate */
    unsigned group_start, group_end;  /* the selector group [start, end) of sorted gate indices */
    unsigned index;                   /* position in the sorted gate list = the selector's value on this gate's rows */
} vpbs_gate;
#define VPBS_UNUSED_SELECTOR 0xFFFFFFFFu
/* Fills the derived fields with the standard parameters of CircuitConfig::standard_recursion_config (135 wires, 80
 * routed, 2 constants) when p0 == 0:  *_from_config constructors.  */
int vpbs_gate_default_params(vpbs_gate* gate);
/* CircuitBuilder::build's gate ordering + selectors.rs selector_polynomials: sorts `gates` by (degree, id), groups them
 * greedily so that group size + gate degree <= max_degree, assigns selector_index / group / index.  max_degree is
 * quotient_degree_factor + 1 (9 under standard_recursion_config), as CircuitBuilder::build passes it.  Outputs the number
 * of selector polynomials and max over gates of num_constraints (CommonCircuitData::num_gate_constraints). */
int vpbs_gates_layout(vpbs_gate* gates, unsigned n_gates, unsigned max_degree, unsigned* num_selectors,
                      unsigned* num_gate_constraints);
/* Gate::id() (the Debug string plonky2 sorts by); returns the length or < 0 */
int vpbs_gate_id(const vpbs_gate* gate, char* buf, size_t len);
/* = evaluate_gate_constraints_base_batch folded with the alphas: for every point x of the LDE coset (leaf order) and
 * every challenge a, d_out[a][x] = sum_i alpha_a^i * sum_g filter_g(x) * constraint_{g,i}(x).  Reads the committed LDEs in
 * HBM: selectors = constants_sigmas columns [0, num_selectors), gate constants = columns [num_selectors, ..).
 * d_out: device, [num_challenges][local LDE length of the batches].  Feed it to vpbs_quotient_permutation. */
int vpbs_gate_terms(vpbs_ctx* ctx, vpbs_batch* constants_sigmas, vpbs_batch* wires, const vpbs_gate* gates, unsigned n_gates,
                    unsigned num_selectors, const uint64_t public_inputs_hash[4], const uint64_t* alphas,
                    unsigned num_challenges, uint64_t* d_out);
/* the same sum at one extension point (verifier side, host only, gates/gate.rs eval_filtered): constants [..][2] are the
 * openings of the constants columns (selectors first), wires [..][2] the wire openings; out [num_challenges][2] */
int vpbs_gate_terms_at(const vpbs_gate* gates, unsigned n_gates, unsigned num_selectors, const uint64_t* constants_at,
                       unsigned n_constants, const uint64_t* wires_at, unsigned n_wires, const uint64_t public_inputs_hash[4],
                       const uint64_t* alphas, unsigned num_challenges, uint64_t* out);
/* ---- witness generation (host only; iop/generator.rs, iop/witness.rs, plonk/permutation_argument.rs) ----
 * Witness rows (SimpleGenerator::run_once of each gate): given the gate's free
 * inputs already present in `row` ([num_wires], one trace row) fills the wires the gate's generators own (outputs,
 * S-box inputs, limbs, intermediate accumulators ...).  constants: the gate constants of that row. */
int vpbs_gate_fill_row(const vpbs_gate* gate, const uint64_t* constants, uint64_t* row);
/* Gadget-level generators (the ones the reference's own gadgets instantiate outside the recursive verifier: builder.is_equal ->
 * EqualityGenerator, le_sum -> BaseSumGenerator, split_le -> WireSplitGenerator; /root/reference/src/vtfhe/ivc_based_vpbs.rs:104-107
 * and the decomposition / rotation gadgets).  Targets are wire positions (column * n + row): a virtual target is identified with a
 * wire it is connected to. */
typedef enum {
    VPBS_GEN_EQUALITY = 0, /* gadgets/arithmetic.rs EqualityGenerator: in = [x, y]; out = [equal, inv]: equal = (x == y), inv = (x - y)^-1 or 0 */
    VPBS_GEN_BASE_SUM,     /* gates/base_sum.rs BaseSumGenerator (le_sum): p0 = base B; in = limbs (little endian); out = [sum] */
    VPBS_GEN_WIRE_SPLIT,   /* gadgets/split_join.rs WireSplitGenerator (split_le): p0 = limbs per gate; in = [integer]; out = the BaseSumGate sum wire of
                              every gate: chunk k = bits [k p0, (k+1) p0) of the canonical integer; error if bits remain.  With p0 = 1 this is
                              SplitGenerator (one bit per output) */
    /* the remaining generators of plonky2's recursive verifier gadgets (FRI verifier: div_extension / inverse_extension; range checks): */
    VPBS_GEN_QUOTIENT_EXT, /* gadgets/arithmetic_extension.rs QuotientGeneratorExtension: in = [num0, num1, den0, den1]; out = [q0, q1] = num / den
                              in GF(p^2); error on a zero denominator */
    VPBS_GEN_COPY,         /* iop/generator.rs CopyGenerator: in = [src]; out = [dst] */
    VPBS_GEN_LOW_HIGH      /* gadgets/range_check.rs LowHighGenerator: p0 = n_log; in = [integer]; out = [low, high]: low = the n_log low bits */
} vpbs_generator_kind;
typedef struct {
    unsigned kind, p0;
    const uint32_t* in;
    unsigned n_in;
    const uint32_t* out;
    unsigned n_out;
} vpbs_generator;

/* A circuit over the supported gates, as CircuitBuilder::build leaves it in ProverOnlyCircuitData / CommonCircuitData:
 * the gate instance of every row, the constants columns, and the copy constraints between routed wires. */
typedef struct {
    unsigned log_n, n_wires, n_routed;   /* degree_bits, config.num_wires (135), config.num_routed_wires (80) */
    const vpbs_gate* gates;              /* laid out by vpbs_gates_layout */
    unsigned n_gates, num_selectors;
    const uint32_t* row_gate;            /* [n]: index into gates of the gate instance on each row */
    const uint64_t* constants;           /* [n_constants_cols][n] column-major: selector columns, then the gate constants */
    unsigned n_constants_cols;
    const uint32_t* copies;              /* [n_copies][2]: wire positions (column * n + row, column < n_routed) constrained equal */
    size_t n_copies;
    const vpbs_generator* generators;    /* gadget-level generators (may be NULL / 0) */
    size_t n_generators;
} vpbs_circuit;
/* gates/selectors.rs selector_polynomials: out [num_selectors][n] = the gate's index on its selector, UNUSED elsewhere */
int vpbs_selector_columns(const vpbs_circuit* circuit, uint64_t* out);
/* WirePartition::get_sigma_polys: sigma values [n_routed][n]; every copy-constraint class becomes one cycle (members in
 * (row, column) order), everything else maps to itself: sigma[col][row] = k_col' * w^row' of the image position */
int vpbs_sigma_values(const vpbs_circuit* circuit, uint64_t* out);
/* generate_partial_witness + full_witness: starting from the preset targets (PartialWitness: positions column * n + row and
 * values) runs every gate generator whose watched wires are set, propagating values through the copy-constraint classes,
 * until all have run.  wires_out: [n_wires][n] (unset wires are 0).  Errors (VPBS_ERR_INVALID, message in err): a class set
 * twice with different values, generators that could not run, a value that does not fit its gate. */
int vpbs_generate_witness(const vpbs_circuit* circuit, const uint32_t* preset_pos, const uint64_t* preset_val, size_t n_preset,
                          uint64_t* wires_out, char* err, size_t err_len);
/* The same, compiled: which generator can run when depends only on the circuit and on WHICH targets the PartialWitness sets, never
 * on their values, so the readiness loop of generate_partial_witness is run once and recorded as a straight-line schedule over one
 * value slot per copy-constraint class.  vpbs_witness_plan_run then replays it for one PartialWitness (values in the order of
 * preset_pos given at creation) and writes the full witness with `threads` host threads (0 = default).  A plan is immutable and may be
 * run from several host threads at once -- the step circuit is proven n + 2 times per PBS (ivc_based_vpbs.rs:302,333,364) with one
 * plan.  Creation fails like vpbs_generate_witness when generators cannot run; a run fails on value errors (a class set twice with
 * different values, a value that does not fit its gate). */
typedef struct vpbs_witness_plan vpbs_witness_plan;
int vpbs_witness_plan_create(const vpbs_circuit* circuit, const uint32_t* preset_pos, size_t n_preset, vpbs_witness_plan** out,
                             char* err, size_t err_len);
int vpbs_witness_plan_run(const vpbs_witness_plan* plan, const uint64_t* preset_val, unsigned threads, uint64_t* wires_out /* [n_wires][n] */,
                          char* err, size_t err_len);
void vpbs_witness_plan_free(vpbs_witness_plan* plan);
/* Two-phase runs, for a chain in which part of the PartialWitness arrives late -- the previous proof of an IVC step
 * (/root/reference/src/vtfhe/ivc_based_vpbs.rs:314-330: set_proof_with_pis_target(&inner_cyclic_proof_with_pis, &proof)): everything that does
 * not depend on the late targets is generated while the previous proof is still being computed.
 *   split     : late[i] != 0 marks preset i (plan order) as late; every generator that reads a late value, directly or not, becomes late.
 *               late[i] = 1, 2, .. k (k <= 16) splits the late phase into STAGES: preset i arrives in stage late[i], a late generator belongs
 *               to the highest stage among what it reads, and stage s can run (vpbs_witness_plan_run_late_stage) as soon as the presets of
 *               stages <= s exist -- an IVC host runs the in-circuit verifier's transcript and vanishing check on the caps and openings of the
 *               previous proof while that proof's FRI stage is still on the device (vpbs_step_inputs.on_section)
 *   run_early : the early presets (the late entries of preset_val are ignored), the early generators, the whole wire matrix (late wires 0)
 *   run_late  : the late presets, the late generators, the late wires written into the same matrix; consumes the state
 * run_early + run_late produce exactly the matrix of vpbs_witness_plan_run.
 * Threads: each phase runs on a pool of host threads that belongs to the plan (created at the phase's first run -- `threads`, 0 = default
 * by host size, or VPBS_EARLY_THREADS / VPBS_LATE_THREADS -- and kept until the plan is freed): generators are grouped by dependency level,
 * wide levels are shared, long hash chains run as lanes beside them.  run_early and run_late of one plan may run concurrently with each
 * other; a second concurrent run of the SAME phase finds the pool taken and runs on its calling thread alone. */
typedef struct vpbs_witness_state vpbs_witness_state;
/* The CPUs this process may use for those pools.  Default: the hardware threads capped by the scheduler affinity and the cgroup CPU quota.
 * A launcher that starts one prover process per GPU of a node gives each its share (bench.py: quota / ranks): every process otherwise sizes
 * its pools for the whole machine.  0 restores the default.  Takes effect for pools created afterwards (a phase's pool keeps the size of its
 * first run).  vpbs_host_cpu_budget returns the figure in force. */
int vpbs_host_set_cpu_budget(unsigned cpus);
unsigned vpbs_host_cpu_budget(void);
/* Threads of the LATE phase's pool for plans split afterwards (0 = default: half the CPU budget, at most 8; the environment variable
 * VPBS_LATE_THREADS overrides both).  The last late stage of an in-circuit verifier is 28 independent FRI queries: a host that runs ONE chain
 * and has 16 CPUs sets 14 (two queries per thread: 11.5 instead of 12.1 ms per chained step); with several chains per GPU the default is the
 * better setting.  Results never depend on it. */
int vpbs_host_set_late_threads(unsigned threads);
/* Threads of the EARLY phase's pool for pools created afterwards (0 = default as above; VPBS_EARLY_THREADS overrides both; an explicit
 * `threads` argument of run_early wins).  The early phase of a chained step runs AHEAD of the step's proof: with c chains per GPU a chain has
 * c proof times to finish it in, and a pool spins between the phase's hundreds of dependency levels.  A host that runs four chains or more
 * per process sets 1 (measured with eight chains on 16 CPUs: the same 7.6 ms per chained proof with 1, 2, 4 or 8 threads, 42 instead of 78
 * CPU-ms per proof; tools/experiments/early_threads_ab.sh); one chain alone keeps the default.  Results never depend on it. */
int vpbs_host_set_early_threads(unsigned threads);
/* How the library's host threads wait for the device.  0: hipStreamSynchronize (spins: lowest latency, one CPU per waiting thread -- a chain's
 * proving thread waits most of the time); 1: blocking (the thread polls the stream and sleeps in between: a wait ends up to ~50 us late,
 * next to no CPU while waiting); -1: default = the environment variable VPBS_BLOCKING_SYNC, else AUTO: blocking when the process may use fewer
 * than 8 CPUs (that is what vpbs_host_blocking_sync reports), and above that for every wait that finds more threads waiting for the device
 * than a quarter of the CPUs -- one chain spins for its latency, eight chains on ten CPUs sleep.  Process-wide; returns the mode in force (vpbs_host_blocking_sync: the same without changing anything).  Measured with four chains per GPU on 2 CPUs: the spinning proving threads alone took both
 * CPUs (tools/prove_ivc.py VPBS_CPU_BY_ROLE). */
int vpbs_host_set_blocking_sync(int on);
int vpbs_host_blocking_sync(void);
/* What a waiting thread looks at.  1 (default): a completion word -- a one-thread kernel behind the work writes a sequence number into host
 * memory the device has mapped and the thread reads that word (spinning, or napping in blocking mode); the HIP runtime is not asked.
 * 0: hipStreamSynchronize / hipStreamQuery.  Either of those makes the HSA runtime's event thread busy-wait for as long as the device has
 * work -- one more CPU per process (measured: 8.5 ms of CPU per 8.9 ms of device work, tools/experiments/graph_cpu_probe.hip).  -1: default =
 * the environment variable VPBS_SYNC_WORD, else 1.  Process-wide; returns the mode in force.  Results never depend on it. */
int vpbs_host_set_sync_word(int on);
int vpbs_witness_plan_split(vpbs_witness_plan* plan, const uint8_t* late /* [n_preset] */, char* err, size_t err_len);
int vpbs_witness_plan_run_early(const vpbs_witness_plan* plan, const uint64_t* preset_val, unsigned threads, uint64_t* wires_out,
                                vpbs_witness_state** state_out, char* err, size_t err_len);
/* run_early into a matrix that still holds what an earlier run_early (+ run_late) of THIS plan left there -- an IVC host cycles a few pinned
 * matrices: only the wire positions that carry values are rewritten; the others, zero since that earlier run, are left alone (zeroing the
 * matrix is a third of the early phase at the paper's parameters).  Anything else in wires_out gives a wrong witness (which the prover's
 * output then fails to verify); the first use of a buffer must go through vpbs_witness_plan_run_early. */
int vpbs_witness_plan_run_early_recycled(const vpbs_witness_plan* plan, const uint64_t* preset_val, unsigned threads, uint64_t* wires_out,
                                         vpbs_witness_state** state_out, char* err, size_t err_len);
int vpbs_witness_plan_run_late(const vpbs_witness_plan* plan, vpbs_witness_state* state, const uint64_t* preset_val, uint64_t* wires_out,
                               char* err, size_t err_len);   /* the state is consumed whether the run succeeds or not */
void vpbs_witness_state_free(vpbs_witness_state* state);   /* only for a state that never reached run_late */
/* Stages of the late phase (1 unless vpbs_witness_plan_split was given stage numbers; 0 before the split).  run_late_stage runs ONE stage
 * ahead of run_late: the presets of that stage are read from preset_val (entries of other stages are not touched), its generators run on
 * the late pool; stages run once each, in ascending order; the state is NOT consumed.  vpbs_witness_plan_run_late[_packed] then runs
 * whatever stages are left and writes the late wires -- the result is the same matrix whichever stages ran ahead.  The late wire positions
 * (vpbs_witness_plan_late_positions) are ordered by stage, so with packed_out a stage leaves its share of the packed values in place at once
 * and run_late_packed (same buffer) only adds the shares of the stages it runs itself.  A failing stage (a
 * value of the proof section that contradicts the circuit) returns VPBS_ERR_INVALID with the message; the state then only goes to
 * vpbs_witness_state_free or run_late (which fails the same way). */
unsigned vpbs_witness_plan_late_stages(const vpbs_witness_plan* plan);
int vpbs_witness_plan_run_late_stage(const vpbs_witness_plan* plan, vpbs_witness_state* state, unsigned stage, const uint64_t* preset_val,
                                     uint64_t* packed_out /* NULL, or the [late_count] buffer the later run_late_packed gets */,
                                     char* err, size_t err_len);
/* The late phase without the matrix: the values of the late wire positions, packed in the order of vpbs_witness_plan_late_positions
 * (count = vpbs_witness_plan_late_count; positions column * n + row, fixed once the plan is split).  For a host whose early matrix is
 * already on the device: upload `values_out` (a few MB instead of the row range of every column) and let vpbs_device_scatter put the
 * words in place.  The host matrix of the early phase is not touched.  Consumes the state like vpbs_witness_plan_run_late. */
int vpbs_witness_plan_run_late_packed(const vpbs_witness_plan* plan, vpbs_witness_state* state, const uint64_t* preset_val,
                                      uint64_t* values_out /* [late_count] */, char* err, size_t err_len);
/* The late phase on its own, for a host whose EARLY phase ran elsewhere (on the device, in batches: vpbs_witness_device_create_early): the
 * early-known values the late phase touches -- what its generators read, what they write as comparers -- are `late_input_count` values, one
 * per copy class, listed by a wire position each (vpbs_witness_plan_late_input_positions: column * n + row); a state seeded with them
 * (vpbs_witness_state_from_late_inputs, values in that order, canonical) is what vpbs_witness_plan_run_late[_packed] then consumes, exactly
 * as if vpbs_witness_plan_run_early had produced it. */
size_t vpbs_witness_plan_late_input_count(const vpbs_witness_plan* plan);   /* 0 before the split */
int vpbs_witness_plan_late_input_positions(const vpbs_witness_plan* plan, uint32_t* out /* [late_input_count] */);
int vpbs_witness_state_from_late_inputs(const vpbs_witness_plan* plan, const uint64_t* values /* [late_input_count] */,
                                        vpbs_witness_state** state_out);
size_t vpbs_witness_plan_late_count(const vpbs_witness_plan* plan);   /* 0 before the split */
int vpbs_witness_plan_late_positions(const vpbs_witness_plan* plan, uint32_t* out /* [late_count] */);
/* out = {row_lo, row_hi}: every wire position run_late writes lies in rows [row_lo, row_hi) -- what has to be uploaded again when the
 * matrix run_early produced is already on the device.  (0, 0) for a plan without late wires. */
int vpbs_witness_plan_late_rows(const vpbs_witness_plan* plan, size_t out[2]);
/* out: {value slots (copy-constraint classes that carry a value), scheduled generators, dependency levels of the device schedule
 * (0: the plan has no device form, see vpbs_witness_device_create), wire positions written by full_witness} */
int vpbs_witness_plan_stats(const vpbs_witness_plan* plan, uint64_t out[4]);
/* The same schedule on the device, for a batch of PartialWitnesses of one circuit (the n + 2 step witnesses of a PBS are independent
 * once vpbs_pbs_accumulator_chain has produced the accumulators): the plan's generators are grouped by dependency level and replayed
 * for `batch` instances at once, values in HBM as [slot][batch]; the level launches are captured in a hipGraph per batch size (the
 * Poseidon-only tail of the schedule is one launch; ~60 ms per run for the step circuit at the paper's parameters, any batch up to 730).  The
 * wires of an instance are then gathered into a device [n_wires][n] matrix that vpbs_prove_step takes with inputs_on_device = 1 --
 * they never cross PCIe.  Every supported gate and gadget generator has a device form (ArithmeticGate operations, bit splits and
 * PoseidonGate rows -- the step circuit's bulk -- have dedicated kernels, the rest run the host's generator code per row).
 * preset_val: host [n_preset][batch] (order of preset_pos at plan creation, instances innermost).  Value errors of any instance (a
 * class set twice with different values, an integer that does not fit, a non-boolean swap) fail the run.  The plan must outlive the
 * device object.  run / wires / read of one object are serialised internally (they share the context's stream and memory pool), so
 * several prover threads may gather their instances from the same object; give every object its own context if its runs should
 * overlap with other work of the same process. */
typedef struct vpbs_witness_device vpbs_witness_device;
int vpbs_witness_device_create(vpbs_ctx* ctx, const vpbs_witness_plan* plan, unsigned max_batch, vpbs_witness_device** out);
/* The EARLY phase of a split plan alone (vpbs_witness_plan_split), for a chain whose steps take the previous proof as a late input: the
 * early parts of a batch of steps are generated here ahead of the chain -- every step's early presets must be known up front, i.e. the
 * host computes the chain's public inputs natively (vpbs_pbs_accumulator_chain, vpbs_hash_chain) -- the late presets' entries of
 * preset_val are ignored.  Per step the host then gathers the instance's wires into the device matrix (vpbs_witness_device_wires: late
 * positions hold zeros), reads back the early values its late phase needs (vpbs_witness_device_read_late_inputs, the order of
 * vpbs_witness_plan_late_input_positions) and runs vpbs_witness_state_from_late_inputs + vpbs_witness_plan_run_late_packed +
 * vpbs_device_scatter: the early phase costs no host CPU and its 70 MB matrix never crosses PCIe. */
int vpbs_witness_device_create_early(vpbs_ctx* ctx, const vpbs_witness_plan* plan, unsigned max_batch, vpbs_witness_device** out);
int vpbs_witness_device_read_late_inputs(vpbs_witness_device* dev, unsigned instance, uint64_t* out /* host [late_input_count] */);
/* The late phase of ONE instance of the last batch, on the device, on top of the early values the object holds for it: preset_val [n_preset]
 * in the plan's order, of which only the late entries (the previous proof's words) are read.  A late value that conflicts with what the
 * circuit fixes -- a proof that does not verify in circuit -- fails the call (vpbs_last_error names the class).  Afterwards
 * vpbs_witness_device_wires gathers the complete witness of the instance. */
int vpbs_witness_device_run_late(vpbs_witness_device* dev, unsigned instance, const uint64_t* preset_val);
/* 1 when the object carries a device schedule of the late phase (an early-only object of a split plan whose late generators all have a
 * device form), 0 otherwise: vpbs_witness_device_run_late is only valid on such an object */
int vpbs_witness_device_has_late(const vpbs_witness_device* dev);
int vpbs_witness_device_run(vpbs_witness_device* dev, const uint64_t* preset_val, unsigned batch);
/* gathers instance `instance` of the last run into d_wires (device, [n_wires][n], fully written) */
int vpbs_witness_device_wires(vpbs_witness_device* dev, unsigned instance, uint64_t* d_wires);
/* values of `count` wire positions (column * n + row) of one instance -> host (e.g. the public inputs) */
int vpbs_witness_device_read(vpbs_witness_device* dev, unsigned instance, const uint32_t* positions, size_t count, uint64_t* out);
void vpbs_witness_device_free(vpbs_witness_device* dev);
/* Checks a complete witness against the circuit on the host: every row satisfies the constraints of its gate (evaluated on the
 * trace values themselves, i.e. on the subgroup) and every copy constraint holds.  Returns 1 = satisfied, 0 = violated (err names the
 * first violation: row, gate, constraint index or the two wire positions), < 0 = malformed arguments.  This is the integration aid
 * for a witness produced elsewhere (e.g. by the reference's Rust generators): if it passes here, the quotient computed by
 * vpbs_prove_step is a polynomial and the proof will verify. */
int vpbs_check_witness(const vpbs_circuit* circuit, const uint64_t* wires /* [n_wires][n] */, const uint64_t public_inputs_hash[4],
                       char* err, size_t err_len);

/* ---- one step proof minus the host-only stages (transcript order of Appendix A.3) ---- */
/* Progress of a step proof, for a host that starts consuming the proof before it is complete (the IVC chain: the next step's in-circuit
 * verifier, ivc_based_vpbs.rs:323-353): fn(user, section) runs on the proving thread, between two launches, when a SECTION of the proof is
 * final in the caller's output buffers --
 *   1              : caps_out (all three caps) and openings_out                      (the FRI stage starts now)
 *   2 .. 1 + R     : fri_out's commit-phase cap of reduction round section - 2           (R = n_rounds of the FRI parameters)
 *   2 + R          : final polynomial and proof-of-work witness, at their final offsets at the end of fri_out
 *                    (the query rounds in between are still missing: they arrive with the return of the call)
 * It must return quickly (hand the work to another thread): the device idles while it runs. */
typedef void (*vpbs_step_section_fn)(void* user, int section);
typedef struct {
    unsigned log_n;                   /* degree_bits: 16 for N=1024, 13 for N=8 (ivc_based_vpbs.rs:54-61 pads to 2^15 / 2^12 gates BEFORE build()) */
    unsigned n_wires;                 /* 135 */
    unsigned n_zs_partial_products;   /* 20: [Z_0, Z_1, pp...] */
    unsigned n_quotient;              /* 16 */
    unsigned num_challenges;          /* 2 */
    int inputs_on_device;             /* 1: the three pointers below are device pointers */
    const uint64_t* wires_values;     /* [n_wires][n]   -> from_values */
    const uint64_t* zs_pp_values;     /* [n_zs_pp][n]   -> from_values; NULL: computed on the device from the wires, the
                                         sigma values below and the transcript's betas/gammas (vpbs_partial_products) */
    const uint64_t* quotient_coeffs;  /* [n_quotient][n]-> from_coeffs; NULL: the quotient chunks are computed on the device
                                         from the committed LDEs with the permutation-argument constraints only
                                         (vpbs_quotient_permutation; needs sigmas via constants_sigmas + n_constants) */
    vpbs_batch* constants_sigmas;     /* committed once per circuit (prover_data.constants_sigmas_commitment) */
    uint64_t circuit_digest[4];
    const uint64_t* public_inputs;    /* host */
    size_t n_public_inputs;
    uint64_t forced_pow;              /* VPBS_POW_ANY or a nonce */
    /* used only when zs_pp_values == NULL (same host/device residency as the other matrices): */
    const uint64_t* sigmas_values;    /* [n_routed][n] sigma polynomial values on H (prover_data.sigmas, column-major) */
    unsigned n_routed;                /* 80: config.num_routed_wires */
    unsigned quotient_degree_factor;  /* 8: chunk size of the partial products */
    unsigned n_constants;             /* leading columns of constants_sigmas that are not sigmas (quotient on device) */
    /* gate constraints of the circuit (quotient on device only): NULL / 0 = none (permutation argument only) */
    const vpbs_gate* gates;           /* laid out by vpbs_gates_layout */
    unsigned n_gates;
    unsigned num_selectors;           /* leading constants columns that are selector polynomials */
    int sigmas_on_device;             /* 1: sigmas_values is a device pointer even when inputs_on_device == 0 -- the sigma values are
                                         circuit data, uploaded once, while the wires of each proof arrive from the host */
    vpbs_step_section_fn on_section;  /* NULL (zero-initialised struct): no progress calls */
    void* on_section_user;
} vpbs_step_inputs;

/* Collectives for a step proof sharded over the GPUs of one node: supplied by the host, so the library
 * stays free of any communication dependency (bench.py / tests: torch.distributed = RCCL over xGMI or gloo; a C++ host
 * would call rccl directly).  Payloads are tiny: 2^cap_height hashes per commitment, ~30 KB of query records per proof. */
typedef int (*vpbs_allgather_fn)(void* user, const uint64_t* local, size_t local_words, uint64_t* full /* [world][local_words] */);
typedef int (*vpbs_allreduce_sum_fn)(void* user, uint64_t* inout, size_t words); /* element-wise wrapping u64 sum */
/* device-resident all-gather: the library has written `local_words` words to d_stage_local (its stream is synchronised);
 * the callback must leave rank r's words at d_stage_full + r * local_words on every rank and return after the data is
 * visible to later work on any stream (e.g. ncclAllGather + stream synchronise). */
typedef int (*vpbs_allgather_dev_fn)(void* user, size_t local_words);
typedef struct {
    unsigned rank, world;               /* world: power of two <= 2^rate_bits */
    vpbs_allgather_fn allgather;
    vpbs_allreduce_sum_fn allreduce_sum;
    void* user;
    /* optional (needed only when the quotient is evaluated on the device in a sharded step): */
    vpbs_allgather_dev_fn allgather_dev;
    uint64_t* d_stage_local;            /* device buffer, stage_capacity_words words */
    uint64_t* d_stage_full;             /* device buffer, world * stage_capacity_words words */
    size_t stage_capacity_words;
} vpbs_comm;

/* The same collectives natively: RCCL over xGMI, bound with dlopen at the first call (no link-time dependency; 0 from
 * vpbs_rccl_available when no librccl.so can be loaded).  Cap hashes, query records and the quotient values move between device buffers
 * with ncclAllGather / ncclAllReduce on the context's stream.  Rank 0 makes the 128-byte id (ncclGetUniqueId) and the host hands it to
 * the other ranks (torch.distributed broadcast in bench.py, the launcher's channel in a C++ / Rust host); every rank of the node then
 * creates its communicator.  stage_words > 0: device staging for the on-device quotient (nc * local LDE length words per rank).
 * VPBS_RCCL_LIB (environment, read once): the library to bind instead of the process's / the system's librccl.so -- the only candidate then.
 * A collective that gets no answer within VPBS_COMM_TIMEOUT_S seconds (default 60) aborts the communicator (ncclCommAbort) and fails; every
 * later call on it fails at once.  After that the context's stream may still hold the copies queued around the aborted collective:
 * vpbs_comm_rccl_destroy gives them a bounded wait, and a stream that does not drain marks the context unusable (vpbs_last_error says so).
 * The rank must then exit non-zero -- a process that has touched the GPU is never restarted in place. */
int vpbs_rccl_available(void);
int vpbs_rccl_unique_id(uint8_t id_out[128]);
int vpbs_comm_rccl_create(vpbs_ctx* ctx, const uint8_t unique_id[128], unsigned rank, unsigned world, size_t stage_words, vpbs_comm* out);
void vpbs_comm_rccl_destroy(vpbs_comm* comm);

/* sizes of the outputs of vpbs_prove_step, in u64 words */
typedef struct {
    size_t cap_words;       /* per cap: 4 << cap_height */
    size_t openings_words;  /* 2 * (n_cs + n_wires + n_zs_pp + n_quotient + num_challenges) */
    size_t fri_words;
} vpbs_step_sizes;
int vpbs_step_sizes_get(const vpbs_ctx* ctx, const vpbs_step_inputs* in, vpbs_step_sizes* out);

/* Runs: wires commit -> observe digest, PI hash, cap -> betas, gammas -> Z/pp commit -> alphas -> quotient commit ->
 * zeta -> openings -> observe -> prove_openings/fri_proof.  The Z / partial-product values and the quotient chunks are either
 * supplied (zs_pp_values / quotient_coeffs) or computed on the device (NULL: vpbs_partial_products; gate constraints of in->gates +
 * permutation argument, vpbs_gate_terms + vpbs_quotient_permutation).
 * caps_out: [3][cap_words] (wires, zs_partial_products, quotient); openings_out: ext values in plonky2 field order
 * constants, plonk_sigmas, wires, plonk_zs, partial_products, quotient_polys (all at zeta) then plonk_zs_next (g*zeta);
 * challenges_out (optional, may be NULL): betas[nc], gammas[nc], alphas[nc], zeta[2]. */
int vpbs_prove_step(vpbs_ctx* ctx, const vpbs_step_inputs* in, uint64_t* caps_out, uint64_t* openings_out,
                    uint64_t* fri_out, vpbs_challenger_state* challenger_out, uint64_t* challenges_out);
/* The same step proof with every commitment coset-sharded over comm->world ranks (one GPU each).  Every rank holds the
 * full input matrices and runs the identical transcript; a rank computes the LDE, leaf hashes and Merkle subtrees of its
 * own cosets only (1/world of the dominant work), caps are assembled with comm->allgather, the FRI rounds are computed
 * redundantly on every rank (cheaper than exchanging them), and the query openings of the sharded oracles are answered
 * by the owning rank and merged with comm->allreduce_sum.  With quotient_coeffs == NULL every rank evaluates the quotient
 * values of its own cosets (the next-row access stays inside a coset), the values are all-gathered on the device
 * (comm->allgather_dev, 8n * 16 B per step: 8 MiB at degree 2^16) and the cheap size-8n iNTT is replicated.  in->constants_sigmas must be a batch committed with
 * vpbs_commit_sharded_dev(.., comm->rank, comm->world, ..) (or an unsharded one).  Every rank returns the complete,
 * identical proof -- bit-identical to vpbs_prove_step on one GPU. */
int vpbs_prove_step_sharded(vpbs_ctx* ctx, const vpbs_step_inputs* in, const vpbs_comm* comm, uint64_t* caps_out,
                            uint64_t* openings_out, uint64_t* fri_out, vpbs_challenger_state* challenger_out,
                            uint64_t* challenges_out);
/* Failure semantics of a sharded step (the reference unwraps every prover error and dies, ivc_based_vpbs.rs:308,339,370; ranks of a node must
 * not hang instead).  Every host collective of vpbs_prove_step_sharded carries one more word per rank, its status: a rank that fails between
 * two collectives still takes part in all the collectives the step has left -- with zeros and its error code -- and returns its own error;
 * every other rank sees the non-zero status at its next collective, does the same, and returns VPBS_ERR_PEER.  No rank returns a proof, no
 * rank waits for a peer that has gone.  (A peer that DIES is the communicator's business: the RCCL communicator of this library gives up
 * after VPBS_COMM_TIMEOUT_S seconds (default 60) and aborts itself; torch.distributed groups take a timeout at creation.)
 * vpbs_prove_step_sharded_fail is for a rank that cannot even start the step (its witness generation failed): it takes part in the step's
 * collectives on the failing side, so that the peers' vpbs_prove_step_sharded calls return VPBS_ERR_PEER.  `in` gives the shape only.
 * vpbs_comm_allgather_checked is the same idea for a host's own all-gathers around the library (the cap of a sharded constants / sigmas
 * commitment): local_status != 0 on any rank makes it return VPBS_ERR_PEER (the failing rank: its status) on every rank. */
int vpbs_prove_step_sharded_fail(vpbs_ctx* ctx, const vpbs_step_inputs* in, const vpbs_comm* comm, int status);
int vpbs_comm_allgather_checked(const vpbs_comm* comm, const uint64_t* local, size_t local_words, uint64_t* full, int local_status);
/* ProofWithPublicInputs::to_bytes layout (util/serialization); returns bytes written or <0.
 * n_constants: how many leading columns of constants_sigmas are `constants` (the rest are plonk_sigmas).  The public-input prefix follows
 * the context's compat table. */
long vpbs_step_proof_to_bytes(const vpbs_ctx* ctx, const vpbs_step_inputs* in, unsigned n_constants,
                              const uint64_t* caps, const uint64_t* openings, const uint64_t* fri, uint8_t* out,
                              size_t out_capacity);

/* = plonk/prover.rs all_wires_permutation_partial_products: Z polynomials and partial products of
 * the permutation argument for every challenge.  wires [>= n_routed][n] and sigmas [n_routed][n] are values on the
 * subgroup (column-major); k_is[j] = 7^j.  out: [num_challenges * (num_prods + 1)][n], num_prods =
 * ceil(n_routed / max_degree) - 1, in the prover's batch order: Z_0..Z_{nc-1}, then the partial products of challenge
 * 0, 1, ...  on_device: all three matrix pointers are device pointers.  VPBS_ERR_INVALID if a denominator is zero. */
int vpbs_partial_products(vpbs_ctx* ctx, const uint64_t* wires, const uint64_t* sigmas, int on_device, unsigned n_routed,
                          unsigned log_n, const uint64_t* betas, const uint64_t* gammas, unsigned num_challenges,
                          unsigned max_degree, uint64_t* out);

/* = plonk/prover.rs compute_quotient_polys for the permutation-argument constraints: the vanishing
 * terms L_0(x)(Z_c(x) - 1) and the partial-product checks, folded with every alpha, plus optional pre-folded gate terms,
 * divided by Z_H on the coset 7<w_8n>, coset-iFFT'd and split into 8 chunks of n coefficients per challenge.
 * All inputs are device-resident committed batches (nothing is downloaded): sigma columns are columns
 * [n_constants, n_constants + n_routed) of constants_sigmas, the routed wires are the first n_routed columns of `wires`.
 * d_gate_terms: NULL or device [num_challenges][8n] in leaf order: sum_g gate_g(x) alpha^g per challenge (multiplied here
 * by alpha^(n_perm_terms)).  out: [num_challenges * 8][n] coefficients, host (out_on_device = 0) or device. */
int vpbs_quotient_permutation(vpbs_ctx* ctx, vpbs_batch* constants_sigmas, unsigned n_constants, vpbs_batch* wires,
                              vpbs_batch* zs_partial_products, unsigned n_routed, const uint64_t* betas,
                              const uint64_t* gammas, const uint64_t* alphas, unsigned num_challenges, unsigned max_degree,
                              const uint64_t* d_gate_terms, uint64_t* out, int out_on_device);

/* ---- verifier (host only; plonky2 plonk/verifier.rs `verify` -> fri/verifier.rs `verify_fri_proof`; the reference calls it
 *      as cd.verify(proof) at /root/reference/src/vtfhe/ivc_based_vpbs.rs:443-447).  Replays the transcript of
 *      vpbs_prove_step, checks proof-of-work, every Merkle path, the alpha-combination, the arity-16 folds and the final
 *      polynomial, and vanishing(zeta) == Z_H(zeta) * t(zeta): the permutation-argument constraints + the circuit's gate
 *      constraints (and with them the PublicInputGate binding) evaluated at zeta from the openings.  That full check is the DEFAULT
 *      (a zero-initialised struct): it needs n_constants, n_routed, quotient_degree_factor and the gates.  fri_only = 1 skips the
 *      vanishing identity -- then "accepted" only says that the openings belong to committed low-degree polynomials, NOT that any
 *      constraint holds (that mode is for proofs over synthetic columns, e.g. bench.py's random traces). */
typedef struct {
    unsigned log_n, rate_bits, cap_height;
    unsigned n_constants_sigmas, n_wires, n_zs_partial_products, n_quotient, num_challenges;
    const uint64_t* constants_sigmas_cap;   /* verifier_only.constants_sigmas_cap: [2^cap_height][4] */
    uint64_t circuit_digest[4];
    const uint64_t* public_inputs;
    size_t n_public_inputs;
    int fri_only;                           /* 0 (default): full verification; 1: transcript + PoW + Merkle paths + FRI only (unsound as a verdict on the circuit) */
    unsigned n_constants, n_routed, quotient_degree_factor;
    const uint64_t* gate_terms_zeta;        /* [num_challenges][2] or NULL (ignored when gates != NULL) */
    const vpbs_gate* gates;                 /* the circuit's gates: their constraints are evaluated at zeta from the openings */
    unsigned n_gates, num_selectors;
    const vpbs_compat* compat;              /* NULL (zero-initialised struct) = vpbs_compat_default; read by vpbs_verify_step (FRI combination),
                                               vpbs_step_proof_from_bytes (public-input prefix) and vpbs_verify_pbs */
} vpbs_verify_inputs;
/* The inverse of vpbs_step_proof_to_bytes: ProofWithPublicInputs bytes -> caps [3][cap], openings, fri (the arrays vpbs_verify_step
 * takes; sizes as vpbs_step_sizes_get reports) and the public inputs.  The shape is taken from `in` (log_n, rate_bits, cap_height, column
 * counts, num_challenges, n_constants); returns the number of public inputs, or < 0 for bytes of another shape (wrong length, wrong
 * Merkle-path lengths, a non-canonical field element, more public inputs than public_inputs_capacity). */
long vpbs_step_proof_from_bytes(const vpbs_verify_inputs* in, const uint8_t* bytes, size_t len, uint64_t* caps, uint64_t* openings,
                                uint64_t* fri, uint64_t* public_inputs_out, size_t public_inputs_capacity);
/* returns 1 = proof accepted, 0 = rejected, < 0 = malformed arguments */
int vpbs_verify_step(const vpbs_verify_inputs* in, const uint64_t* caps /* [3][cap] */, const uint64_t* openings,
                     const uint64_t* fri);

/* ---- one verifiable PBS as one call: the IVC chain of verified_pbs (/root/reference/src/vtfhe/ivc_based_vpbs.rs:159-386) ----
 * The cyclic step circuit and its dummy circuit arrive as data (what CircuitBuilder::build leaves behind: vpbs_circuit + the PartialWitness
 * targets in the order verified_pbs sets them + the public-input targets).  vpbs_ivc_create commits their constants / sigmas, derives the
 * verifier data, compiles and splits the witness plans and allocates the wire matrices; vpbs_ivc_prove_pbs then runs
 *     base proof (cyclic_base_proof: a proof of the dummy circuit whose public inputs carry the initial values) -> n + 2 step proofs of the cyclic circuit, each taking the previous proof as a witness
 * with the early witness phase of the next step and its upload running on two host threads beside the proof of the current one, and
 * returns the LAST proof serialised (ProofWithPublicInputs bytes): the input of vpbs_verify_pbs.  One chain at a time per vpbs_ivc; several
 * vpbs_ivc objects (one context each) run side by side.
 *   preset_pos of the cyclic circuit: previous proof's words [proof_words] | its public inputs [n_pi] | condition | GGSW [ggsw_len] | mask |
 *                                     own verifier data [4 + cap] | dummy verifier data [4 + cap] | the dummy circuit's proof [proof_words] |
 *                                     its public inputs [n_pi]                                          (wire positions column * n + row)
 *                                     -- the last three are what plonky2's DummyProofGenerator fills (recursion/dummy_circuit.rs): the SECOND
 *                                     proof slot of conditionally_verify_cyclic_proof_or_dummy, every word of which the circuit selects
 *                                     against the first by `condition`; vpbs_ivc_create proves the dummy circuit once (all-zero public
 *                                     inputs) and presets its proof in every step
 *   preset_pos of the dummy circuit : its public inputs [n_pi];  n_pi = 2 K N + 9 + 4 + cap words
 *   bsk [n_lwe][ggsw_len] NTT domain (vpbs_keygen's layout), ksk [ggsw_len], ct [n_lwe + 1], testv [N]: host arrays
 *   steps: 0 or n_lwe + 2 = the whole chain; fewer = a prefix (tests)
 * vpbs_ivc_prove_pbs returns the number of proof bytes, or < 0 (err: which step failed and why). */
typedef struct {
    const vpbs_circuit* circuit;
    const uint32_t* preset_pos;
    size_t n_preset;
    const uint32_t* pi_pos;     /* public-input targets of the cyclic circuit; never read for the dummy circuit (may be NULL there) */
    size_t n_pi;                /* cyclic circuit: 2 K N + 9 + 4 + cap words; dummy circuit: the same number or 0 (its public inputs are its PartialWitness) */
    size_t proof_words;         /* cyclic circuit: words of one proof in target order (caps, openings, FRI); 0 for the dummy circuit */
} vpbs_ivc_circuit;
typedef struct vpbs_ivc vpbs_ivc;
typedef struct {
    double seconds;             /* base proof + steps, wall clock */
    unsigned steps;
    double base_proof_ms, late_witness_ms, late_rows_upload_ms, prove_step_ms, early_witness_ms;   /* per step, except the base proof */
    double late_ahead_ms;       /* per step: late witness stages that ran on a second thread WHILE the previous proof's FRI stage was on the
                                   device (not part of late_witness_ms, which is what is left on the critical path after the proof) */
} vpbs_ivc_timing;
/* comm: NULL = one GPU.  Otherwise (BASELINE config 4) every rank of the node calls with its own context and its vpbs_comm (callbacks or
 * vpbs_comm_rccl_create): the chain is sequential, so the GPUs share every STEP -- constants / sigmas committed with
 * vpbs_commit_sharded_dev, every step proven with vpbs_prove_step_sharded -- while every rank generates the (identical) witnesses on its
 * host and ends with the identical proof.  The communicator must outlive the vpbs_ivc. */
int vpbs_ivc_create(vpbs_ctx* ctx, const vpbs_ivc_circuit* cyclic, const vpbs_ivc_circuit* dummy, unsigned N, unsigned K, size_t ggsw_len,
                    const vpbs_comm* comm, vpbs_ivc** out, char* err, size_t err_len);
void vpbs_ivc_free(vpbs_ivc* ivc);
/* Progress hook: fn(user, done) is called on the thread inside vpbs_ivc_prove_pbs with done = 0 when the base proof exists and with
 * done = 1 .. steps after each chained step proof (its words are on the host, nothing of that step is left on the device).  A host places
 * its clock with it -- bench.py times K chained steps after W warm-up steps between done == W and the return -- or reports progress.  The
 * early witness phases of later steps keep running on their own threads while the hook runs (that pipelining is the design, not warm-up).
 * NULL removes the hook. */
/* The early witness phases on the DEVICE: batch > 0 switches vpbs_ivc_prove_pbs to a pipeline in which the host runs only the late phase
 * (the in-circuit verifier's rows, which need the previous proof).  What a step's early phase needs of its predecessor are the
 * predecessor's public inputs, and those are known without proving anything -- accumulators from vpbs_pbs_accumulator_chain (hence ELL and
 * LOGB here), the two chain hashes from the native sponge (one host thread), counter, verifier data -- so the early phases of `batch`
 * consecutive steps are generated at once by two early-only device objects on contexts of their own (vpbs_witness_device_create_early),
 * gathered per step into the matrix the prover reads, and only the early values the late phase touches come back to the host.  The 70 MB
 * matrix of a step no longer crosses PCIe and a chain needs about one host CPU instead of five; the proofs are the same bytes.  A batch of
 * 64 costs 2 x 0.9 GB of device memory at the paper's parameters (1.78 M value slots per instance).  batch = 0 returns to the host pipeline. */
int vpbs_ivc_set_device_witness(vpbs_ivc* ivc, unsigned ELL, unsigned LOGB, unsigned batch, int late_on_device);
/* late_on_device != 0: the late phase on the device as well -- once the previous proof exists its words go to the device object that holds
 * the step's early values, the late generators run there as a level schedule of their own (vpbs_witness_device_run_late: ~160 dependent
 * levels, one instance), and all the step's wires are gathered into the prover's matrix: the host generates no witness at all (it keeps
 * the transcript, the native hash chains and the launches).  For ranks with very few CPUs; with CPUs to spare the host's late phase
 * (eight threads, AVX-512 Poseidon) is the faster one. */
typedef void (*vpbs_ivc_step_fn)(void* user, unsigned done);
int vpbs_ivc_set_step_callback(vpbs_ivc* ivc, vpbs_ivc_step_fn fn, void* user);
/* text of the last failure of a vpbs_ivc_set_* call on this object ("" when none); valid until the next call on the object */
const char* vpbs_ivc_last_error(const vpbs_ivc* ivc);
/* circuit digest [4] then constants/sigmas cap of either circuit (what a verifier of the chain holds); either pointer may be NULL */
int vpbs_ivc_verifier_data(const vpbs_ivc* ivc, uint64_t* cyclic_vk, uint64_t* dummy_vk);
long vpbs_ivc_prove_pbs(vpbs_ivc* ivc, const uint64_t* testv, const uint64_t* ct, const uint64_t* bsk, const uint64_t* ksk, unsigned n_lwe,
                        unsigned steps, uint8_t* proof_out, size_t capacity, vpbs_ivc_timing* timing, char* err, size_t err_len);

/* = verify_pbs (/root/reference/src/vtfhe/ivc_based_vpbs.rs:388-489): the statement of ONE verifiable PBS checked on the last proof of its IVC
 * chain, in the reference's order -- claimed test vector, counter = n + 2, output ciphertext = the proof's accumulator, cd.verify(proof)
 * (full vpbs_verify_step), check_cyclic_proof_verifier_data (the trailing public inputs are this circuit's digest and cap), the hash chain
 * over [dummy GGSW, bsk_0 .. bsk_{n-1}, ksk] and the one over [ct[n], ct[0] .. ct[n-1], 0].  Host only.
 * circuit: shape, cap, digest and gates of the cyclic step circuit as for vpbs_verify_step (its public_inputs fields are ignored; the
 * public inputs come out of the proof bytes: acc_init [K N] | counter | accumulator [K N] | key hash [4] | LWE hash [4] | digest [4] | cap).
 * Returns 1 = accepted, 0 = rejected (why: the first failing check), < 0 = malformed arguments. */
typedef struct {
    const vpbs_verify_inputs* circuit;
    unsigned N, K, n_lwe;
    size_t ggsw_len;          /* K ELL K N: words of one flattened GGSW */
    const uint64_t* testv;    /* [N] */
    const uint64_t* out_ct;   /* [K][N] the bootstrapped ciphertext the caller holds; required (the reference asserts it, :440-442) */
    const uint64_t* ct;       /* [n_lwe + 1] the LWE input */
    const uint64_t* bsk;      /* [n_lwe][ggsw_len] NTT domain, Ggsw::flatten order */
    const uint64_t* ksk;      /* [ggsw_len] */
} vpbs_verify_pbs_inputs;
int vpbs_verify_pbs(const vpbs_verify_pbs_inputs* in, const uint8_t* proof_bytes, size_t len, char* why, size_t why_len);

/* ---- kernel-level entry points (host buffers; used by parity tests and by callers outside the prover) ---- */
int vpbs_k_poseidon_batch(vpbs_ctx* ctx, uint64_t* states /* [n][12] in place */, size_t n);
int vpbs_k_hash_rows(vpbs_ctx* ctx, const uint64_t* rows /* [n][len] */, size_t n, unsigned len, uint64_t* out /* [n][4] */);
int vpbs_k_intt(vpbs_ctx* ctx, const uint64_t* values, unsigned ncols, unsigned log_n, uint64_t* coeffs_out);
/* out[c][j] = poly_c(shift * w^{bitrev(j)}), j < n << rate_bits (plonky2 leaf order) */
int vpbs_k_coset_lde(vpbs_ctx* ctx, const uint64_t* coeffs, unsigned ncols, unsigned log_n, unsigned rate_bits,
                     uint64_t shift, uint64_t* out);
/* MerkleTree::new(leaves, cap_height) over row-major leaves [n_leaves][leaf_len]: cap_out [2^cap_height][4] */
int vpbs_k_merkle_cap(vpbs_ctx* ctx, const uint64_t* leaves, size_t n_leaves, unsigned leaf_len, unsigned cap_height,
                      uint64_t* cap_out);
/* reference negacyclic NTT (/root/reference/src/vtfhe/crypto/poly.rs:27-64), batched, in place on [batch][1<<log_n] */
int vpbs_k_negacyclic_ntt(vpbs_ctx* ctx, uint64_t* data, unsigned batch, unsigned log_n, int inverse);
/* the params_{N}.rs tables (ROOTS, INVROOTS, NINV) regenerated per /root/reference/src/ntt/gen_param_file.sage */
/* n Poseidon permutations on the HOST, in place ([n][12], canonical in and out): eight at a time on AVX-512 lanes where the CPU has them
 * (returns 1), one after the other otherwise (returns 0).  The batched form is what the witness generator and the verifier use for
 * independent permutations; exported for parity tests. */
int vpbs_k_poseidon_host(uint64_t* states, size_t n);
/* shader clock (MHz) of one CU measured over 20 us on the context's stream, after everything queued before it (s_memtime / s_memrealtime) */
int vpbs_k_clock_probe(vpbs_ctx* ctx, double* mhz_out);
int vpbs_ntt_params(unsigned log_n, uint64_t* roots, uint64_t* invroots, uint64_t* ninv);

/* ---- native TFHE data path of a step (witness-generation core) ----
 * One step of the verifiable PBS exactly as the step circuit computes it (/root/reference/src/vtfhe/ivc_based_vpbs.rs:99-125):
 *   first_step : acc_out = rotate(acc_in, -mask)                                  (the LWE body)
 *   otherwise  : x = last_step ? acc_in : rotate(acc_in, mask) - acc_in           (mod.rs:80-117: mod switch to 2N, rounded)
 *                acc_out = GGSW (x) x  [+ acc_in unless last_step]                 (ggsw_ct.rs:98-112, glev_ct.rs:92-110,
 *                signed base-2^LOGB decomposition glwe_poly.rs:28-50, top ELL limbs, negacyclic NTT crypto/poly.rs:9-64)
 * batched over independent accumulators.  acc: [batch][K][N]; masks: [batch]; ggsw: NTT-domain bootstrapping-key element in
 * the order of Ggsw::flatten() = [K glevs][ELL glwes][K polys][N] (crypto/ggsw.rs:61-63), one for all instances or
 * (ggsw_per_instance) [batch] of them; all canonical field elements; host or device pointers (on_device). */
typedef struct {
    unsigned log_N; /* ring dimension N = 2^log_N (<= 2048) */
    unsigned K;     /* GLWE dimension + 1 */
    unsigned ELL;   /* decomposition levels kept */
    unsigned LOGB;  /* log2 of the decomposition base */
} vpbs_tfhe_params;
int vpbs_blind_rotate_step(vpbs_ctx* ctx, const vpbs_tfhe_params* params, unsigned batch, const uint64_t* acc_in,
                           const uint64_t* masks, const uint64_t* ggsw, int ggsw_per_instance, int first_step, int last_step,
                           uint64_t* acc_out, int on_device);

/* The whole accumulator chain of one PBS, as verified_pbs drives it (/root/reference/src/vtfhe/ivc_based_vpbs.rs:280-371):
 *   step 0      first_step with mask = lwe_ct[n] (the body), acc_init = (0, .., 0, testv)
 *   step 1..n   CMUX with bsk[x], mask = lwe_ct[x]
 *   step n+1    last_step with the key-switching key (mask 0)
 * lwe_ct: [n + 1]; bsk: [n][K*ELL*K*N] (NTT domain, Ggsw::flatten order); ksk: [K*ELL*K*N].  accs_out receives every
 * intermediate accumulator, [n + 2][K][N] -- the `current accumulator` public inputs of the n + 2 step proofs (host memory). */
int vpbs_pbs_accumulator_chain(vpbs_ctx* ctx, const vpbs_tfhe_params* params, unsigned n, const uint64_t* acc_init,
                               const uint64_t* lwe_ct, const uint64_t* bsk, const uint64_t* ksk, uint64_t* accs_out);

/* ---- seeded key / ciphertext generation (the reference draws all of it from unseeded RNGs,
 *      /root/reference/src/main.rs:40-52, crypto/poly.rs:72-88, crypto/lwe.rs:12,43,55) ----
 * Everything is a function of ONE seed (generator: csrc/keygen.hip header; restated by tests/tfhe_oracle.py):
 *   s_to   [K][N]      Glwe::partial_key(n_lwe)   (crypto/glwe.rs:19-40): binary, only the leading n_lwe coefficients non-zero
 *   s_lwe  [n_lwe]     flatten_partial_key        = those coefficients
 *   s_glwe [K-1][N]    Glwe::key_gen              (crypto/glwe.rs:15-17)
 *   bsk    [n_lwe][K][ELL][K][N]   compute_bsk(s_lwe, s_glwe, sigma_glwe) (crypto/mod.rs:29-45): Ggsw::encrypt(s_glwe, constant(s_i)).ntt_forward(),
 *                                  NTT domain, Ggsw::flatten order -- what vpbs_pbs_accumulator_chain / the step circuit's GGSW targets take
 *   ksk    [K][ELL][K][N]          Ggsw::compute_ksk(s_to, s_glwe, sigma_lwe) (crypto/ggsw.rs:38-48)
 * Noise: a rounded Gaussian stand-in with standard deviation floor(sigma q + 1/2) in integer arithmetic (Irwin-Hall of twelve 32-bit
 * uniforms), so that host, device and oracle agree bit for bit.  The GGSWs are generated on the device (728 x 16 GLWE encryptions at the
 * paper's parameters); bsk / ksk are host pointers, or device pointers when keys_on_device != 0; the three key outputs are host arrays.
 * Any output may be NULL. */
typedef struct {
    unsigned log_N, K, ELL, LOGB; /* as vpbs_tfhe_params */
    unsigned n_lwe;               /* LWE dimension (728 in /root/reference/src/main.rs:27) */
    uint64_t seed;
    double sigma_glwe, sigma_lwe; /* relative to q, as main.rs:29-30 (4.99027217501041e-8, 1.17021618159313e-5) */
} vpbs_keygen_params;
int vpbs_keygen(vpbs_ctx* ctx, const vpbs_keygen_params* params, uint64_t* s_lwe, uint64_t* s_glwe, uint64_t* s_to, uint64_t* bsk,
                uint64_t* ksk, int keys_on_device);
/* lwe::encrypt(s_lwe, message, sigma_lwe) (crypto/lwe.rs:55-64) with the mask / noise streams of `nonce` (< 2^24); host only; ct: [n_lwe + 1] */
int vpbs_lwe_encrypt(const vpbs_keygen_params* params, const uint64_t* s_lwe, uint64_t message, uint64_t nonce, uint64_t* ct);
/* get_testv(p, get_delta(2 p)) (crypto/mod.rs:17-27, crypto/lwe.rs:50-52; main.rs:47-48): testv [N] (may be NULL), *delta */
int vpbs_testv(unsigned log_N, unsigned p, uint64_t* testv, uint64_t* delta);
/* Glwe::decrypt (crypto/glwe.rs:60-63): m = body - sum_j a_j s_j over X^N + 1; s: [K-1][N], ct: [K][N] coefficient domain, host arrays */
int vpbs_glwe_decrypt(vpbs_ctx* ctx, unsigned log_N, unsigned K, const uint64_t* s, const uint64_t* ct, uint64_t* m_out);

/* ---- memory helpers for hosts that do not link the HIP runtime themselves (a Rust or plain C++ caller) ----
 * pinned host memory (hipHostMalloc): witness matrices written there reach the device at PCIe speed (70.8 MB in 1.3 ms instead of ~15 ms
 * from pageable memory); device buffers for per-circuit data that is uploaded once (the sigma values of vpbs_step_inputs.sigmas_values with
 * sigmas_on_device = 1, resident wire matrices with inputs_on_device = 1). */
void* vpbs_host_alloc(size_t bytes);
void vpbs_host_free(void* p);
int vpbs_device_alloc(vpbs_ctx* ctx, size_t words, uint64_t** out);
int vpbs_device_upload(vpbs_ctx* ctx, uint64_t* d_dst, const uint64_t* host_src, size_t words); /* returns after the copy has completed */
/* The same copy on the context's upload stream, for a second host thread: it may run while another call (vpbs_prove_step ...) is in
 * progress on the context, touches none of the context's state and reports failures by return code only (vpbs_last_error is not set).
 * An IVC host uploads the early-phase wires of the NEXT step with it while the current step is being proven. */
int vpbs_device_upload_bg(vpbs_ctx* ctx, uint64_t* d_dst, const uint64_t* host_src, size_t words);
/* Rows [row_lo, row_hi) of every column of a column-major [n_cols][n] matrix (one strided copy; the late phase of a split witness plan
 * only changes the rows vpbs_witness_plan_late_rows reports).  Returns after the copy has completed. */
int vpbs_device_upload_rows(vpbs_ctx* ctx, uint64_t* d_dst, const uint64_t* host_src, unsigned n_cols, size_t n, size_t row_lo, size_t row_hi);
/* d_dst[positions[i]] = host_values[i], i < count: the packed values of a late witness phase (vpbs_witness_plan_run_late_packed) put in
 * place in the device-resident matrix of the early phase.  d_positions: the uint32 positions of vpbs_witness_plan_late_positions, uploaded
 * once (vpbs_device_alloc of (count + 1) / 2 words + vpbs_device_upload); d_stage: count words of device scratch; host_values should be
 * pinned (vpbs_host_alloc).  One copy of 8 count bytes + one kernel on the context's stream; returns when both have completed. */
int vpbs_device_scatter(vpbs_ctx* ctx, uint64_t* d_dst, const uint64_t* d_positions, const uint64_t* host_values, size_t count,
                        uint64_t* d_stage);
void vpbs_device_free(vpbs_ctx* ctx, uint64_t* d_ptr);

/* ---- per-kernel device timing (HIP events on the ctx stream) ---- */
/* on: 0 off, 1 every kernel group, 2 only the dominant kernel (leaf_hash) */
int vpbs_timing_enable(vpbs_ctx* ctx, int on);
/* writes a JSON object {"kernel": {"ms": total, "count": launches}, ...} and resets the accumulators */
int vpbs_timing_report(vpbs_ctx* ctx, char* buf, size_t len);
/* The shader clock the chip sustained UNDER the dominant kernel while timing was on: one wave of every leaf-hash launch measures its own
 * lifetime in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime); average over the launches since the last call (at most 1024).
 * A loaded MI355X runs well below the boost clock an idle probe reads, and the issue-rate figures of bench.py are priced at THIS clock. */
int vpbs_timing_shader_clock(vpbs_ctx* ctx, double* mhz_out, unsigned* samples_out);

#ifdef __cplusplus
}
#endif
#endif
