// Launch wrappers of the gfx950 kernels (one HIP stream per prover context; no hidden synchronisation).
// Buffers are device pointers unless stated; field elements are canonical u64.
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "../../include/vpbs_prover.h"

namespace vpbs {
using gl::u32;
using gl::u64;

// ---------- ntt.hip ----------
// roots[j] = w^j, j < n, w = primitive n-th root (or its inverse for the inverse transform)
// roots: root_table_words(log_n) words -- the n powers of w (w^-1 for inverse) followed by the transform kernels' own tables for the PLAIN
// transform of that direction (ntt.hip: from 2^12 points on the block twiddles tau^k of the radix-16 rounds, [round][k][block], with 1/n folded
// into the first round of the inverse)
size_t root_table_words(unsigned log_n);
void launch_root_table(hipStream_t s, u64* roots, unsigned log_n, bool inverse);
// prescale[r][i] = (shift * w_{log_n+rate_bits}^r)^i, r < 2^rate_bits, i < n  (a plain table of powers: the quotient's unshift table,
// and the LDE table of transforms below 2^12 points)
void launch_prescale_table(hipStream_t s, u64* table, unsigned log_n, unsigned rate_bits, u64 shift);
// the table launch_coset_lde wants for (log_n, rate_bits, shift): from 2^12 points on the block twiddles of every coset shift * w_big^r
// (the shift is part of the twiddles: no prescale pass), [coset][round][k][block]; below that the prescale table
size_t lde_table_words(unsigned log_n, unsigned rate_bits);
void launch_lde_table(hipStream_t s, u64* table, unsigned log_n, unsigned rate_bits, u64 shift);
// values -> coefficients, natural order in and out (PolynomialValues::ifft), batched over columns.
// scratch: [ncols][n] device words (may alias nothing); in/out column-major.
void launch_intt(hipStream_t s, const u64* values, u64* coeffs, u64* scratch, const u64* inv_roots, unsigned ncols,
                 unsigned log_n);
// coefficients -> coset LDE in leaf order: out[c][brev_rate(r)*n + q] = value at natural index r + (brev_logn(q) << rate)
// i.e. out[c][j] = poly_c(shift * w_{big}^{brev_big(j)})   (PolynomialCoeffs::lde + coset_fft + reverse_index_bits)
// block_first / n_blocks: compute only the leaf blocks [block_first, block_first + n_blocks) of the 2^rate_bits blocks
// (a block = one coset = n consecutive leaves); out is then [ncols][n_blocks * n].  n_blocks = 0 means all blocks.
// lde_table: launch_lde_table's for the same (log_n, rate_bits, shift)
void launch_coset_lde(hipStream_t s, const u64* coeffs, u64* out, const u64* roots, const u64* lde_table, unsigned ncols,
                      unsigned log_n, unsigned rate_bits, unsigned block_first = 0, unsigned n_blocks = 0);
// batched negacyclic NTT of the reference (crypto/poly.rs:9-64): in place, [batch][n]; roots = ROOTS/INVROOTS table
void launch_negacyclic(hipStream_t s, u64* data, const u64* table, unsigned batch, unsigned log_n, bool inverse, u64 ninv);

// ---------- permutation.hip ----------
// Z and partial products (all_wires_permutation_partial_products): out [nc * (num_prods + 1)][n] in batch order
// (Z_0..Z_{nc-1}, then the partial products of challenge 0, 1, ..); wires [>= n_routed][n], sigmas [n_routed][n] values
// on H; roots = forward root table of size n; scratch: nc * (n + ceil(n/256) + n_chunks * n) words; *d_zero_flag |= 1 on a zero
// denominator.  d_betas / d_gammas: device arrays of nc elements.
void launch_partial_products(hipStream_t s, const u64* wires, const u64* sigmas, const u64* roots, unsigned n_routed, unsigned log_n,
                             unsigned max_degree, const u64* d_betas, const u64* d_gammas, unsigned num_challenges, u64* out,
                             u64* scratch, unsigned* d_zero_flag);

// ---------- quotient.hip ----------
// compute_quotient_polys restricted to the permutation argument (+ optional alpha-folded gate terms, leaf order, [nc][8n]).
// *_lde: committed LDE columns (column stride 8n, leaf order): routed wires, sigmas, Z/partial products (batch order).
// roots_big / inv_roots_big: root tables of size 8n; unshift_table[i] = 7^-i (i < 8n); d_apow: [nc][n_terms + 1] powers of the
// alphas (n_terms = nc * (1 + ceil(n_routed / max_degree))); betas/gammas: host arrays; q_leaf, q_nat: [nc][8n] scratch;
// out_coeffs: [nc * 2^rate_bits][n] quotient chunks.  nc <= 4, rate_bits <= 3.
// l0[j] = L_0(7 w^brev(j)) for the 8n coset points in leaf order (computed once per degree)
void launch_l0_table(hipStream_t s, const u64* roots_big, unsigned log_n, unsigned rate_bits, u64* l0);
// phase 1: quotient values q[a][j_local] for the local leaves [leaf_offset, leaf_offset + local_len) (column stride of the
// *_lde arrays = local_len; gate terms, if any, in the same local leaf order [nc][local_len]); l0_table is the full table.
void launch_quotient_values(hipStream_t s, const u64* wires_lde, const u64* sigmas_lde, const u64* zs_pp_lde, const u64* roots_big,
                            const u64* l0_table, const u64* d_gate_terms, const u64* d_apow, const u64* betas, const u64* gammas,
                            unsigned n_routed, unsigned log_n, unsigned rate_bits, unsigned max_degree, unsigned nc, size_t leaf_offset,
                            size_t local_len, u64* q_leaf_local, bool raw = false);
// raw = true leaves out the gate terms and the division by Z_H; this joins them in afterwards:
// q <- (q + alpha_a^(n_terms) (g0 + g1 + g2)) / Z_H   (g*: [nc][local_len] gate-term lanes, null = unused; apow_last: host [nc])
void launch_quotient_combine_planes(hipStream_t s, u64* q_local, const u64* planes, unsigned n_planes, const u64* apow_last, unsigned log_n,
                                    unsigned rate_bits, unsigned nc, size_t leaf_offset, size_t local_len);
void launch_quotient_combine(hipStream_t s, u64* q_local, const u64* g0, const u64* g1, const u64* g2, const u64* apow_last, unsigned log_n,
                             unsigned rate_bits, unsigned nc, size_t leaf_offset, size_t local_len);
// phase 2: q_gathered is rank-major [world][nc][local_len] (world * local_len = 8n); q_nat, scratch: [nc][8n]
void launch_quotient_finish(hipStream_t s, const u64* q_gathered, size_t local_len, const u64* inv_roots_big, const u64* unshift_table,
                            unsigned log_n, unsigned rate_bits, unsigned nc, u64* q_nat, u64* scratch, u64* out_coeffs);

// Launch heuristics of one context (vpbs_ctx_set_option; the environment variables of the same names are only the DEFAULTS a context is
// created with: VPBS_WIDE_THRESHOLD, VPBS_MERKLE_CLIMB, VPBS_GATES_FUSED, VPBS_GATE_ITEMS, VPBS_GATE_LANES)
struct Tuning {
    size_t wide_threshold = (size_t)1 << 14;   // launches with at most this many independent permutations use the 16-lane Poseidon form
    size_t fri_leaf_wide_threshold = (size_t)1 << 14;   // the same for the FRI round leaves (several dependent permutations per leaf)
    bool merkle_climb = true;                  // the latency-bound upper levels of a tree in fused multi-level launches
    bool gates_fused = true;                   // all gate constraints in one launch (false: one launch per gate type)
    unsigned gate_items = 5;                   // reserved (VPBS_OPT_GATE_ITEMS: the kernel it tuned was removed in round 6); kept so the option reads back
    bool gates_tile = true;                    // the one-launch gate kernel that stages a 64-point tile of every column in LDS
    static Tuning from_env();
};

// ---------- gates.hip ----------
// evaluate_gate_constraints_base_batch folded with the alphas: d_out[a][j] = sum_i alpha_a^i sum_g filter_g(j) c_{g,i}(j) for the
// `len` local leaves of the wires / constants LDEs (column stride len).  d_apow: [nc][pow_stride] powers of the alphas,
// pow_stride >= max num_constraints.  gates: laid out by vpbs_gates_layout (host array).
// lanes (optional): two helper streams + events + two scratch outputs [nc][len]; the gate kernels are then spread over three streams
// (fork / join on s) so that VALU-bound and HBM-bound gates overlap
struct GateLanes {
    hipStream_t stream[2];
    hipEvent_t fork, join[2];
    u64* out[2];
    // optional independent work that joins the lane assignment (the permutation part of the quotient): run(stream, arg)
    void (*extra)(hipStream_t, void*) = nullptr;
    void* extra_arg = nullptr;
    unsigned extra_weight = 0;
    bool skip_sum = false;      // leave the lanes unsummed (the caller combines them)
    bool used[3] = {false, false, false};  // out: which lanes received gate kernels (lane 0 = d_out)
};
void launch_gate_terms(hipStream_t s, const u64* wires_lde, const u64* consts_lde, size_t len, const vpbs_gate* gates, unsigned n_gates,
                       unsigned num_selectors, const u64 pi_hash[4], const u64* d_apow, unsigned pow_stride, unsigned nc, u64* d_out,
                       GateLanes* lanes = nullptr);
// The same sum in ONE launch, the LDS-tile kernel (gates.hip): a workgroup stages 64 points of every column in LDS and its eight waves share the
// gates; it writes one plane d_planes[nc][len] = the value launch_gate_terms would produce.  gate_terms_planes: how many planes the gate set
// needs -- 1, or 0 = not supported by this path (tune.gates_tile off, len not a multiple of 64, a gate set that fits no tile plan): use
// launch_gate_terms.
unsigned gate_terms_planes(const vpbs_gate* gates, unsigned n_gates, unsigned num_selectors, const Tuning& tune, size_t len);
unsigned launch_gate_terms_fused(hipStream_t s, const Tuning& tune, const u64* wires_lde, const u64* consts_lde, size_t len, const vpbs_gate* gates, unsigned n_gates,
                                 unsigned num_selectors, const u64 pi_hash[4], const u64* d_apow, unsigned pow_stride, unsigned nc, u64* d_planes);
void launch_sum_planes(hipStream_t s, const u64* d_planes, unsigned n_planes, size_t words, u64* d_out);
// throws DeviceError(VPBS_ERR_INVALID) unless the gate list fits batches with these column counts
void validate_gates(const vpbs_gate* gates, unsigned n_gates, unsigned num_selectors, unsigned n_constants_cols, unsigned n_wires);
// host, GF(p^2): the same folded sum at one point from openings ([..][2] arrays); out [nc][2]
void gate_terms_at(const vpbs_gate* gates, unsigned n_gates, unsigned num_selectors, const u64* constants_at, unsigned n_constants,
                   const u64* wires_at, unsigned n_wires, const u64 pi_hash[4], const u64* alphas, unsigned nc, u64* out);

// ---------- tfhe.hip ----------
// One step of the verifiable PBS on `batch` independent accumulators (reference ivc_based_vpbs.rs:99-125): acc [batch][K][N],
// masks [batch], ggsw in NTT domain [K][ELL][K][N] (ggsw_instance_stride = 0: shared by all instances) ; roots/invroots: the
// params_{N} tables on the device; limbs_hat: scratch [batch][K][ELL][N].  ELL * N * 8 B of LDS per workgroup.
void launch_blind_rotate_step(hipStream_t s, const u64* acc_in, const u64* masks, const u64* ggsw, size_t ggsw_instance_stride,
                              const u64* roots, const u64* invroots, u64 ninv, unsigned log_n, unsigned K, unsigned ELL, unsigned LOGB,
                              unsigned batch, int first_step, int last_step, u64* limbs_hat, u64* acc_out);

// ---------- hash.hip ----------
// digests[j] = hash_or_noop(leaf j), leaf j = lde[c][j] over c (column-major LDE, leaf-order index)
// clock_sample: nullptr, or two device words that receive {shader cycles, 100 MHz ticks} over the lifetime of one wave of the launch
void launch_leaf_hash(hipStream_t s, const u64* lde, unsigned ncols, size_t n_leaves, size_t col_stride, u64* digests, u64* clock_sample = nullptr);
// FRI round leaves: leaf l = flatten(values[arity*l .. arity*(l+1))) of ext values stored SoA [2][m]
void launch_fri_leaf_hash(hipStream_t s, const Tuning& tune, const u64* v0, const u64* v1, size_t n_leaves, unsigned arity_bits, u64* digests);
// parents[i] = two_to_one(children[2i], children[2i+1])
void launch_merkle_level(hipStream_t s, const Tuning& tune, const u64* children, u64* parents, size_t n_parents);
// builds every level above `level0` up to the cap; levels are stored back to back: level k at digests + off[k]
void launch_merkle_tree(hipStream_t s, const Tuning& tune, u64* digests, const size_t* level_off, unsigned n_levels, size_t n_leaves);
// batch of independent permutations (test hook)
void launch_permute_batch(hipStream_t s, u64* states, size_t n);
// hash_no_pad over rows of a row-major matrix [n][len] (test hook / chain hashing of key material)
void launch_hash_rows(hipStream_t s, const u64* rows, size_t n, unsigned len, u64* out);
// proof-of-work search: smallest nonce in [start, start+count) whose response has >= pow_bits leading zeros.
// state12: duplex state with the pending inputs already overwritten; pos: slot of the nonce. *result = min nonce or ~0
// tune.wide_threshold below its default (a context that shares the GPU): the range in rounds of 2^15 with early exits instead of at once
void launch_pow_search(hipStream_t s, const Tuning& tune, const u64* state12_host, unsigned pos, unsigned pow_bits, u64 start, u64 count,
                       u64* d_result);

// ---------- fri.hip ----------
// power tables of up to 4 points in one launch: out + 2 n t holds z_t^i (ext, AoS [n][2]), t < count
void launch_ext_powers(hipStream_t s, const gl::Ext* points, unsigned count, size_t n, u64* out);
// out[c] = sum_i coeffs[c][i] * zpow[i]   (p.to_extension().eval(z)); out AoS [ncols][2]
void launch_eval_ext(hipStream_t s, const u64* coeffs, unsigned ncols, size_t n, size_t col_stride, const u64* zpow, u64* out);
// the same for up to 8 coefficient matrices ("segments", each with its own point-power table) in ONE pair of launches; results
// for all columns back to back: out [sum ncols][2], followed by scratch for [sum ncols * ceil(n / 4096)][2] partial sums
struct EvalSegments {
    struct Seg {
        const u64* coeffs;
        const u64* zpow;
        size_t col_stride;
        unsigned ncols;
    } seg[8];
    unsigned count;
};
void launch_eval_ext_multi(hipStream_t s, const EvalSegments& segs, size_t n, u64* out);
// F[i] = sum_j alpha^j * poly_j[i]  (ReducingFactor::reduce_polys_base); polys: device array of column pointers;
// alpha_pows AoS [n_polys][2]; F SoA (f0[n], f1[n])
// partial_scratch: 2 * COMBINE_GROUPS * n device words
constexpr unsigned COMBINE_GROUPS = 16;
void launch_combine(hipStream_t s, const u64* const* polys, unsigned n_polys, const u64* alpha_pows, size_t n, u64* f0, u64* f1,
                    u64* partial_scratch);
// final <- final * scale + (F / (X - z))  with the quotient's top coefficient 0 (divide_by_linear + pad);
// zpow/zinvpow: AoS power tables of z and z^-1 of length n; final SoA
// totals_scratch: 2 * ceil(n / 256) device words
void launch_divide_accumulate(hipStream_t s, const u64* f0, const u64* f1, const u64* zpow, const u64* zinvpow, gl::Ext scale,
                              size_t n, u64* fin0, u64* fin1, u64* totals_scratch);
// out = in * X (coefficient shift; the top coefficient of `in` must be 0); in and out must not alias
void launch_shift_up(hipStream_t s, const u64* in0, const u64* in1, u64* out0, u64* out1, size_t n);
// out[i] = sum_{j<arity} in[arity*i + j] * beta^j
void launch_fold(hipStream_t s, const u64* in0, const u64* in1, size_t n_out, unsigned arity_bits, gl::Ext beta, u64* out0, u64* out1);
// Query openings (fri_prover_query_rounds: tree.get + tree.prove for every query and tree) in one launch.
struct OpenTree {
    const u64* data0;      // initial tree: column-major LDE; FRI tree: component-0 values
    const u64* data1;      // FRI tree: component-1 values (nullptr for initial trees)
    const u64* digests;    // level 0 at digests, level k at digests + level_off[k]
    size_t col_stride;     // initial tree: distance between columns
    size_t level_off[24];
    u32 leaf_len;          // ncols, or 2 << arity_bits
    u32 n_siblings;
    u32 index_shift;       // leaf index = x_index >> index_shift
    u32 arity_bits;        // FRI trees only
    size_t out_off;        // word offset of this tree's record inside one query record
    // multi-GPU: this rank fills the record only for leaf indices in [leaf_lo, leaf_hi) (zeros otherwise; the ranks'
    // records are then summed).  Unsharded: [0, all).  data0/digests index with (leaf - leaf_lo).
    size_t leaf_lo, leaf_hi;
};
constexpr unsigned MAX_OPEN_TREES = 12, MAX_QUERIES = 128;
struct OpenArgs {
    OpenTree trees[MAX_OPEN_TREES];
    u64 x_index[MAX_QUERIES];
    u32 n_trees, n_queries;
    size_t record_words;
};
// d_args: device copy of OpenArgs; out: [n_queries][record_words]
void launch_open_queries(hipStream_t s, const OpenArgs* d_args, unsigned n_trees, unsigned n_queries, u64* out);
}  // namespace vpbs
