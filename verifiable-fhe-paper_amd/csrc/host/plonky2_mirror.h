// Host-side mirror of the plonky2 0.2.0 interface around the hot path, in C++ (the reference's Rust toolchain is not
// available in this image).  Same names, argument meaning and failure behaviour as the functions the reference's
// prove() walks through -- /root/reference/src/vtfhe/ivc_based_vpbs.rs:302-308 -> plonk/prover.rs -> fri/oracle.rs:
//   Challenger            iop/challenger.rs
//   PolynomialBatch       fri/oracle.rs       (from_values, from_coeffs, get_lde_values, prove_openings)
//   FriParams/FriConfig   fri/mod.rs, fri/reduction_strategies.rs
//   FriInstanceInfo, FriBatchInfo, FriPolynomialInfo   fri/structure.rs
//   fri_proof, fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds   fri/prover.rs
// Errors surface as vpbs::DeviceError (the reference `.unwrap()`s anyhow errors: ivc_based_vpbs.rs:308,339,370).
#pragma once
#include <array>
#include <cstring>
#include <vector>

#include "../context.h"

namespace plonky2 {
using vpbs::u32;
using vpbs::u64;
using Ext = gl::Ext;
using HashOut = std::array<u64, 4>;
using MerkleCap = std::vector<HashOut>;

// iop/challenger.rs: duplex sponge over Poseidon, overwrite mode, challenges popped from the end of the rate portion
class Challenger {
public:
    vpbs_challenger_state st;
    Challenger() { vpbs_challenger_init(&st); }
    explicit Challenger(const vpbs_challenger_state& s) : st(s) {}
    void observe_element(u64 e) { vpbs_challenger_observe(&st, &e, 1); }
    void observe_elements(const u64* e, size_t n) { vpbs_challenger_observe(&st, e, n); }
    void observe_hash(const HashOut& h) { observe_elements(h.data(), 4); }
    void observe_cap(const u64* cap, size_t n_hashes) { observe_elements(cap, 4 * n_hashes); }
    void observe_extension_element(Ext e) {
        const u64 v[2] = {e.c0, e.c1};
        observe_elements(v, 2);
    }
    u64 get_challenge() { return vpbs_challenger_get(&st); }
    std::vector<u64> get_n_challenges(size_t n) {
        std::vector<u64> v(n);
        for (auto& x : v) x = get_challenge();
        return v;
    }
    Ext get_extension_challenge() {
        const u64 a = get_challenge(), b = get_challenge();
        return Ext{a, b};
    }
};

struct FriConfig {
    unsigned rate_bits = 3, cap_height = 4, proof_of_work_bits = 16, num_query_rounds = 28;
};
struct FriParams {
    FriConfig config;
    bool hiding = false;
    unsigned degree_bits = 0;
    std::vector<unsigned> reduction_arity_bits;
    bool mul_final_by_x = false;
    unsigned lde_bits() const { return degree_bits + config.rate_bits; }
    size_t lde_size() const { return (size_t)1 << lde_bits(); }
    unsigned final_poly_bits() const {
        unsigned d = degree_bits;
        for (unsigned a : reduction_arity_bits) d -= a;
        return d;
    }
    // FriReductionStrategy::ConstantArityBits(4, 5) of standard_recursion_config (fri/reduction_strategies.rs), evaluated with the
    // rate_bits / cap_height the caller's FriConfig carries (the arities depend on both)
    static FriParams standard(unsigned degree_bits, unsigned rate_bits = 3, unsigned cap_height = 4) {
        FriParams p;
        p.config.rate_bits = rate_bits;
        p.config.cap_height = cap_height;
        p.degree_bits = degree_bits;
        unsigned d = degree_bits;
        while (d > 5 && d + rate_bits >= cap_height + 4) {
            p.reduction_arity_bits.push_back(4);
            d -= 4;
        }
        return p;
    }
    // every Merkle tree of the proof (initial trees over the LDE, one per reduction round) must be at least as tall as its cap
    bool caps_fit() const {
        unsigned lg = degree_bits + config.rate_bits;
        if (config.cap_height > lg) return false;
        for (unsigned a : reduction_arity_bits) {
            if (a > lg) return false;
            lg -= a;
            if (config.cap_height > lg) return false;
        }
        return true;
    }
    static FriParams from_c(const vpbs_fri_params& c, unsigned degree_bits) {
        FriParams p;
        p.config = FriConfig{c.rate_bits, c.cap_height, c.pow_bits, c.num_query_rounds};
        p.degree_bits = degree_bits;
        p.reduction_arity_bits.assign(c.arity_bits, c.arity_bits + c.n_rounds);
        p.mul_final_by_x = c.mul_final_by_x != 0;
        return p;
    }
};

struct FriPolynomialInfo {
    u32 oracle_index, polynomial_index;
};
struct FriBatchInfo {
    Ext point;
    std::vector<FriPolynomialInfo> polynomials;
};
struct FriInstanceInfo {
    std::vector<FriBatchInfo> batches;
};

// fri/oracle.rs PolynomialBatch: `polynomials` (coefficients) + `merkle_tree` (LDE leaves + digests), device-resident
class PolynomialBatch {
public:
    vpbs_batch* h = nullptr;
    PolynomialBatch() = default;
    explicit PolynomialBatch(vpbs_batch* b) : h(b) {}
    PolynomialBatch(const PolynomialBatch&) = delete;
    PolynomialBatch& operator=(const PolynomialBatch&) = delete;
    PolynomialBatch(PolynomialBatch&& o) noexcept : h(o.h) { o.h = nullptr; }
    ~PolynomialBatch() { vpbs_batch_free(h); }
    // from_values(values, rate_bits, blinding, cap_height, timing, fft_root_table): rate_bits / cap_height come from
    // the ctx; blinding must be false (zero_knowledge is off in standard_recursion_config)
    // comm != nullptr: coset-sharded commitment (this rank's share)
    static PolynomialBatch from_values(vpbs_ctx* ctx, const u64* d_values, unsigned ncols, unsigned log_n, bool blinding = false,
                                       const vpbs_comm* comm = nullptr) {
        VPBS_REQUIRE(!blinding, "blinding (zero-knowledge salts) is not part of the vPBS configuration");
        return PolynomialBatch(vpbs::commit_device(ctx, d_values, ncols, log_n, true, comm ? comm->rank : 0, comm ? comm->world : 1));
    }
    static PolynomialBatch from_coeffs(vpbs_ctx* ctx, const u64* d_coeffs, unsigned ncols, unsigned log_n, bool blinding = false,
                                       const vpbs_comm* comm = nullptr) {
        VPBS_REQUIRE(!blinding, "blinding (zero-knowledge salts) is not part of the vPBS configuration");
        return PolynomialBatch(vpbs::commit_device(ctx, d_coeffs, ncols, log_n, false, comm ? comm->rank : 0, comm ? comm->world : 1));
    }
    // merkle_tree.cap; for a sharded batch the ranks' cap entries are assembled with comm->allgather
    void merkle_cap(u64* out, const vpbs_comm* comm = nullptr) const {
        if (h->n_shards == 1) {
            vpbs::batch_cap_to_host(h, out);
            return;
        }
        VPBS_REQUIRE(comm && comm->allgather && comm->world == h->n_shards && comm->rank == h->shard, "sharded batch needs its communicator");
        std::vector<u64> local(4 * h->cap_len());
        vpbs::batch_cap_to_host(h, local.data());
        if (comm->allgather(comm->user, local.data(), local.size(), out) != 0)
            throw vpbs::DeviceError{VPBS_ERR_DEVICE, "cap all-gather failed"};
    }
    // prove_openings(instance, oracles, challenger, fri_params, timing) -> FriProof as flat words (vpbs_prover.h)
    // comm: needed when any oracle is sharded (query records are merged with comm->allreduce_sum)
    static void prove_openings(vpbs_ctx* ctx, const FriInstanceInfo& instance, const std::vector<vpbs_batch*>& oracles,
                               Challenger& challenger, const FriParams& fri_params, u64 forced_pow, u64* proof_out,
                               const vpbs_comm* comm = nullptr, vpbs_step_section_fn on_section = nullptr, void* on_section_user = nullptr);
};

size_t fri_proof_words(const FriParams& p, const std::vector<size_t>& ncols);
}  // namespace plonky2
