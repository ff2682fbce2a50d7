// Host orchestration of the opening proof and of one step proof, on top of the gfx950 kernels.
// Mirrors plonky2 0.2.0 fri/oracle.rs `PolynomialBatch::prove_openings`, fri/prover.rs `fri_proof` /
// `fri_committed_trees` / `fri_proof_of_work` / `fri_prover_query_rounds`, and the transcript skeleton of
// plonk/prover.rs `prove` -- the code behind
// /root/reference/src/vtfhe/ivc_based_vpbs.rs:302-308, :333-339, :364-370.
// The Fiat-Shamir transcript is strictly sequential: every cap / opening set comes back to the host (512 B..4 KiB),
// is absorbed by the host Challenger, and the next challenge is passed to the next kernel as an argument.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "host/plonky2_mirror.h"

using vpbs::DeviceError;
using vpbs::Timed;
using vpbs::u32;
using vpbs::u64;

namespace plonky2 {

size_t fri_proof_words(const FriParams& p, const std::vector<size_t>& ncols) {
    const unsigned log_lde = p.lde_bits();
    const size_t cap = (size_t)4 << p.config.cap_height;
    size_t w = p.reduction_arity_bits.size() * cap;
    size_t per_q = 0;
    for (size_t nc : ncols) per_q += nc + 4 * (size_t)(log_lde - p.config.cap_height);
    unsigned lg = log_lde;
    for (unsigned ab : p.reduction_arity_bits) {
        lg -= ab;
        per_q += ((size_t)2 << ab) + 4 * (size_t)(lg - p.config.cap_height);
    }
    w += per_q * p.config.num_query_rounds;
    w += (size_t)2 << p.final_poly_bits();
    return w + 1;
}

namespace {
struct FriRoundTree {
    u64* values = nullptr;   // SoA [2][len], leaf order
    u64* digests = nullptr;
    std::vector<size_t> level_off;
    size_t n_values = 0, n_leaves = 0;
    unsigned arity_bits = 0;
};

// fri_proof_of_work: smallest nonce whose response has >= pow_bits leading zeros (or validate a forced one)
u64 fri_proof_of_work(vpbs_ctx* ctx, Challenger& ch, unsigned pow_bits, u64 forced) {
    auto valid = [&](u64 w) {
        Challenger c = ch;
        c.observe_element(w);
        const u64 r = c.get_challenge();
        return pow_bits == 0 || (r >> (64 - pow_bits)) == 0;
    };
    u64 witness;
    if (forced != VPBS_POW_ANY) {
        if (forced >= gl::P || !valid(forced)) throw DeviceError{VPBS_ERR_POW, "forced proof-of-work nonce is not valid"};
        witness = forced;
    } else {
        // duplex_intermediate_state: sponge with the buffered inputs already written; the nonce goes to slot input_len
        u64 st[12];
        for (int i = 0; i < 12; ++i) st[i] = ch.st.sponge[i];
        for (u32 i = 0; i < ch.st.input_len; ++i) st[i] = ch.st.input[i];
        const unsigned pos = ch.st.input_len;
        u64* d_res = ctx->alloc_words(1);
        witness = VPBS_POW_ANY;
        // expected nonce ~2^pow_bits: a first chunk of 2^(pow_bits+1) candidates hits with probability 1 - e^-2
        const u64 span = (u64)1 << std::min(22u, pow_bits + 1);
        for (u64 start = 0; witness == VPBS_POW_ANY; start += span) {
            if (start > ((u64)1 << 40)) {
                ctx->release(d_res);
                throw DeviceError{VPBS_ERR_POW, "proof-of-work search exhausted"};
            }
            Timed t(ctx, "pow_search");
            VPBS_HIP(hipMemsetAsync(d_res, 0xFF, sizeof(u64), ctx->stream));
            vpbs::launch_pow_search(ctx->stream, ctx->tune, st, pos, pow_bits, start, span, d_res);
            ctx->d2h_sync(&witness, d_res, sizeof(u64));
        }
        ctx->release(d_res);
        if (!valid(witness)) throw DeviceError{VPBS_ERR_POW, "device proof-of-work result failed the host re-check"};
    }
    ch.observe_element(witness);
    (void)ch.get_challenge();  // pow_response
    return witness;
}
}  // namespace

void PolynomialBatch::prove_openings(vpbs_ctx* ctx, const FriInstanceInfo& instance, const std::vector<vpbs_batch*>& oracles,
                                     Challenger& challenger, const FriParams& fp, u64 forced_pow, u64* proof_out,
                                     const vpbs_comm* comm, vpbs_step_section_fn on_section, void* on_section_user) {
    hipStream_t s = ctx->stream;
    const unsigned degree_bits = fp.degree_bits, rate_bits = fp.config.rate_bits, cap_h = fp.config.cap_height;
    const size_t n = (size_t)1 << degree_bits;
    VPBS_REQUIRE(rate_bits == ctx->rate_bits && cap_h == ctx->cap_height, "FRI params disagree with the context");
    VPBS_REQUIRE(fp.config.num_query_rounds <= vpbs::MAX_QUERIES, "too many query rounds");
    VPBS_REQUIRE(oracles.size() + fp.reduction_arity_bits.size() <= vpbs::MAX_OPEN_TREES, "too many trees");
    const bool multi = comm && comm->world > 1;
    for (auto* o : oracles) {
        VPBS_REQUIRE(o && o->log_n == degree_bits && o->ctx == ctx, "oracle does not match degree/context");
        VPBS_REQUIRE(o->n_shards == 1 || (multi && comm->allreduce_sum && o->n_shards == comm->world && o->shard == comm->rank),
                     "sharded oracle without a matching communicator");
    }

    std::vector<void*> scratch;  // released at the end (stream-ordered pool)
    auto words = [&](size_t w) {
        u64* p = ctx->alloc_words(w);
        scratch.push_back(p);
        return p;
    };
    struct Cleanup {
        vpbs_ctx* c;
        std::vector<void*>& v;
        ~Cleanup() {
            (void)vpbs::stream_sync(c->stream);
            for (void* p : v) c->release(p);
        }
    } cleanup{ctx, scratch};

    // ---- alpha-combination: final_poly = sum_i alpha^(k_i) (F_i - F_i(z_i)) / (X - z_i) ----
    const Ext alpha = challenger.get_extension_challenge();
    size_t max_polys = 0;
    for (auto& b : instance.batches) max_polys = std::max(max_polys, b.polynomials.size());
    std::vector<u64> h_apow(2 * max_polys);
    {
        Ext a = gl::ext(1);
        for (size_t j = 0; j < max_polys; ++j) {
            h_apow[2 * j] = a.c0;
            h_apow[2 * j + 1] = a.c1;
            a = gl::mul(a, alpha);
        }
    }
    u64* d_apow = words(2 * max_polys);
    VPBS_HIP(hipMemcpyAsync(d_apow, h_apow.data(), sizeof(u64) * 2 * max_polys, hipMemcpyHostToDevice, s));
    u64* fin = words(2 * n);  // SoA [2][n]
    VPBS_HIP(hipMemsetAsync(fin, 0, sizeof(u64) * 2 * n, s));
    u64* F = words(2 * n);
    // power tables of every opening point and of its inverse, one launch (at most 2 batches = 4 tables per launch)
    const size_t n_b = instance.batches.size();
    u64* ztables = words(4 * n * n_b);  // batch b: z^i at ztables + 4 n b, z^-i at ztables + 4 n b + 2 n
    {
        Timed t(ctx, "fri_divide");
        for (size_t b0 = 0; b0 < n_b; b0 += 2) {
            Ext pts[4];
            unsigned cnt = 0;
            for (size_t b = b0; b < n_b && b < b0 + 2; ++b) {
                VPBS_REQUIRE(!(instance.batches[b].point.c0 == 0 && instance.batches[b].point.c1 == 0), "opening point is zero");
                pts[cnt++] = instance.batches[b].point;
                pts[cnt++] = gl::inv(instance.batches[b].point);
            }
            vpbs::launch_ext_powers(s, pts, cnt, n, ztables + 4 * n * b0);
        }
    }
    u64* div_totals = words(2 * ((n + 255) / 256));
    u64* combine_scratch = words(2 * (size_t)vpbs::COMBINE_GROUPS * n);
    std::vector<std::vector<const u64*>> h_ptrs(instance.batches.size());
    for (size_t b = 0; b < instance.batches.size(); ++b) {
        const FriBatchInfo& bi = instance.batches[b];
        const size_t np = bi.polynomials.size();
        VPBS_REQUIRE(np > 0, "empty opening batch");
        h_ptrs[b].resize(np);
        for (size_t j = 0; j < np; ++j) {
            const auto& pi = bi.polynomials[j];
            VPBS_REQUIRE(pi.oracle_index < oracles.size() && pi.polynomial_index < oracles[pi.oracle_index]->ncols, "bad FriPolynomialInfo");
            h_ptrs[b][j] = oracles[pi.oracle_index]->d_coeffs + (size_t)pi.polynomial_index * n;
        }
        const u64** d_ptrs = reinterpret_cast<const u64**>(words(np));
        VPBS_HIP(hipMemcpyAsync(d_ptrs, h_ptrs[b].data(), sizeof(u64*) * np, hipMemcpyHostToDevice, s));
        {
            Timed t(ctx, "fri_combine");
            vpbs::launch_combine(s, d_ptrs, (unsigned)np, d_apow, n, F, F + n, combine_scratch);
        }
        {
            Timed t(ctx, "fri_divide");
            const u64* zpow = ztables + 4 * n * b;
            const u64* zinvpow = zpow + 2 * n;
            // shift_poly: final *= alpha^count (count = polynomials reduced in this batch), then += quotient
            vpbs::launch_divide_accumulate(s, F, F + n, zpow, zinvpow, gl::pow(alpha, np), n, fin, fin + n, div_totals);
        }
    }
    if (fp.mul_final_by_x) {
        u64* shifted = words(2 * n);
        vpbs::launch_shift_up(s, fin, fin + n, shifted, shifted + n, n);
        fin = shifted;
    }

    // ---- fri_committed_trees ----
    const size_t n_rounds = fp.reduction_arity_bits.size();
    const size_t cap_words = (size_t)4 << cap_h;
    std::vector<FriRoundTree> trees(n_rounds);
    u64* coeffs = fin;  // SoA [2][len]
    unsigned log_len = degree_bits;
    u64 shift = gl::GENERATOR;
    u64* w = proof_out;
    for (size_t r = 0; r < n_rounds; ++r) {
        const unsigned ab = fp.reduction_arity_bits[r];
        const size_t len = (size_t)1 << log_len, lde_len = len << rate_bits;
        VPBS_REQUIRE(log_len >= ab, "arity larger than the polynomial");
        FriRoundTree& t = trees[r];
        t.arity_bits = ab;
        t.n_values = lde_len;
        t.n_leaves = lde_len >> ab;
        t.values = words(2 * lde_len);
        {
            Timed tm(ctx, "fri_lde");
            vpbs::launch_coset_lde(s, coeffs, t.values, ctx->roots(log_len, false), ctx->lde_table(log_len, rate_bits, shift), 2, log_len,
                                   rate_bits);
        }
        const size_t dig_words = vpbs::merkle_layout(t.n_leaves, cap_h, t.level_off);
        t.digests = words(dig_words);
        {
            Timed tm(ctx, "fri_tree");
            vpbs::launch_fri_leaf_hash(s, ctx->tune, t.values, t.values + lde_len, t.n_leaves, ab, t.digests);
            vpbs::launch_merkle_tree(s, ctx->tune, t.digests, t.level_off.data(), (unsigned)t.level_off.size(), t.n_leaves);
        }
        ctx->d2h_sync(w, t.digests + t.level_off.back(), sizeof(u64) * cap_words);
        challenger.observe_cap(w, cap_words / 4);
        w += cap_words;
        if (on_section) on_section(on_section_user, 2 + (int)r);   // this round's cap is final in proof_out
        const Ext beta = challenger.get_extension_challenge();
        const size_t new_len = len >> ab;
        u64* folded = words(2 * new_len);
        {
            Timed tm(ctx, "fri_fold");
            vpbs::launch_fold(s, coeffs, coeffs + len, new_len, ab, beta, folded, folded + new_len);
        }
        coeffs = folded;
        log_len -= ab;
        shift = gl::pow(shift, (u64)1 << ab);
    }
    // final polynomial (the coefficients above len are the ones the reference truncates: they are zero)
    const size_t final_len = (size_t)1 << log_len;
    std::vector<u64> h_final(2 * final_len), final_words(2 * final_len);
    ctx->d2h_sync(h_final.data(), coeffs, sizeof(u64) * 2 * final_len);
    for (size_t i = 0; i < final_len; ++i) {
        final_words[2 * i] = h_final[i];
        final_words[2 * i + 1] = h_final[final_len + i];
    }
    challenger.observe_elements(final_words.data(), final_words.size());

    // ---- fri_proof_of_work ----
    const u64 pow_witness = fri_proof_of_work(ctx, challenger, fp.config.proof_of_work_bits, forced_pow);
    {   // final polynomial and nonce go to their final place now (behind the query rounds), so that a host reading the proof section by
        // section (vpbs_step_inputs.on_section, section 2) finds everything but the query rounds
        std::vector<size_t> ncols;
        for (auto* o : oracles) ncols.push_back(o->ncols);
        u64* tail = proof_out + fri_proof_words(fp, ncols) - 1 - final_words.size();
        std::memcpy(tail, final_words.data(), sizeof(u64) * final_words.size());
        tail[final_words.size()] = pow_witness;
        if (on_section) on_section(on_section_user, 2 + (int)n_rounds);
    }

    // ---- fri_prover_query_rounds ----
    vpbs::OpenArgs args{};
    args.n_queries = fp.config.num_query_rounds;
    const size_t lde_size = fp.lde_size();
    for (unsigned q = 0; q < args.n_queries; ++q) args.x_index[q] = challenger.get_challenge() % lde_size;
    size_t off = 0;
    unsigned nt = 0;
    for (auto* o : oracles) {
        vpbs::OpenTree& t = args.trees[nt++];
        t.data0 = o->d_lde;
        t.data1 = nullptr;
        t.digests = o->d_digests;
        t.col_stride = o->lde_len();
        t.leaf_len = o->ncols;
        t.n_siblings = o->n_levels() - 1;
        t.index_shift = 0;
        // ownership: a sharded oracle answers for its own leaf range; a replicated one is answered by rank 0 only
        t.leaf_lo = o->leaf_offset();
        t.leaf_hi = o->leaf_offset() + o->lde_len();
        if (multi && o->n_shards == 1 && comm->rank != 0) t.leaf_hi = t.leaf_lo = 0;
        for (unsigned k = 0; k < o->n_levels(); ++k) t.level_off[k] = o->level_off[k];
        t.out_off = off;
        off += t.leaf_len + 4 * (size_t)t.n_siblings;
    }
    unsigned total_shift = 0;
    for (size_t r = 0; r < n_rounds; ++r) {
        const FriRoundTree& ft = trees[r];
        total_shift += ft.arity_bits;
        vpbs::OpenTree& t = args.trees[nt++];
        t.data0 = ft.values;
        t.data1 = ft.values + ft.n_values;
        t.digests = ft.digests;
        t.leaf_len = 2u << ft.arity_bits;
        t.n_siblings = (unsigned)ft.level_off.size() - 1;
        t.index_shift = total_shift;
        t.arity_bits = ft.arity_bits;
        t.leaf_lo = 0;
        t.leaf_hi = (multi && comm->rank != 0) ? 0 : ft.n_leaves;  // FRI round trees are replicated: rank 0 answers
        for (size_t k = 0; k < ft.level_off.size(); ++k) t.level_off[k] = ft.level_off[k];
        t.out_off = off;
        off += t.leaf_len + 4 * (size_t)t.n_siblings;
    }
    args.n_trees = nt;
    args.record_words = off;
    auto* d_args = static_cast<vpbs::OpenArgs*>(ctx->alloc_bytes(sizeof args));
    scratch.push_back(d_args);
    u64* d_rec = words(off * args.n_queries);
    VPBS_HIP(hipMemcpyAsync(d_args, &args, sizeof args, hipMemcpyHostToDevice, s));
    {
        Timed tm(ctx, "fri_open_queries");
        vpbs::launch_open_queries(s, d_args, nt, args.n_queries, d_rec);
    }
    ctx->d2h_sync(w, d_rec, sizeof(u64) * off * args.n_queries);
    if (multi && comm->allreduce_sum(comm->user, w, off * args.n_queries) != 0)
        throw DeviceError{VPBS_ERR_DEVICE, "query-record all-reduce failed"};
    w += off * args.n_queries;
    {   // the tail was written right after the proof of work; the two layouts must agree
        std::vector<size_t> ncols;
        for (auto* o : oracles) ncols.push_back(o->ncols);
        VPBS_REQUIRE(w == proof_out + fri_proof_words(fp, ncols) - 1 - final_words.size(), "FRI proof layout: query rounds and word count disagree");
    }
    VPBS_HIP(hipGetLastError());
}
}  // namespace plonky2

namespace {
// all_wires_permutation_partial_products on the device; d_out: [nc * (num_prods + 1)][n]
// deferred (nullable): instead of synchronising for the zero-denominator flag here, the flag travels to pinned memory behind the kernels and
// *deferred points at it -- the caller checks it after ITS next synchronisation (the step proof: with the cap of the Z commitment), so the
// host goes on enqueueing that commitment at once.  *deferred stays null when the flag was checked here.
void partial_products_device(vpbs_ctx* ctx, const u64* d_wires, const u64* d_sigmas, unsigned n_routed, unsigned log_n,
                             const u64* betas, const u64* gammas, unsigned nc, unsigned max_degree, u64* d_out,
                             volatile unsigned** deferred = nullptr) {
    VPBS_REQUIRE(log_n >= 1 && n_routed >= 1 && max_degree >= 1 && nc >= 1, "bad partial-product shape");
    const size_t n = (size_t)1 << log_n;
    hipStream_t s = ctx->stream;
    u64* d_ch = ctx->alloc_words(2 * nc + 1);
    const size_t n_chunks = (n_routed + max_degree - 1) / max_degree;
    u64* scratch = ctx->alloc_words((size_t)nc * (n + (n + 255) / 256 + n_chunks * n));
    std::vector<u64> h(2 * nc + 1, 0);
    for (unsigned c = 0; c < nc; ++c) {
        h[c] = betas[c];
        h[nc + c] = gammas[c];
    }
    unsigned flag = 0;
    try {
        VPBS_HIP(hipMemcpyAsync(d_ch, h.data(), sizeof(u64) * h.size(), hipMemcpyHostToDevice, s));
        {
            Timed t(ctx, "partial_products");
            vpbs::launch_partial_products(s, d_wires, d_sigmas, ctx->roots(log_n, false), n_routed, log_n, max_degree, d_ch, d_ch + nc, nc,
                                          d_out, scratch, reinterpret_cast<unsigned*>(d_ch + 2 * nc));
        }
        if (deferred) *deferred = ctx->d2h_deferred_flag(d_ch + 2 * nc);
        if (!deferred || !*deferred) ctx->d2h_sync(&flag, d_ch + 2 * nc, sizeof(unsigned));
    } catch (...) {
        ctx->release(d_ch);
        ctx->release(scratch);
        throw;
    }
    ctx->release(d_ch);
    ctx->release(scratch);
    VPBS_REQUIRE(flag == 0, "zero denominator in the permutation argument (the reference's batch inverse would panic)");
}

// the one-launch gate kernel (gates.hip) is the default; vpbs_ctx_set_option(VPBS_OPT_GATES_FUSED, 0) selects the per-gate launches (kept
// for comparison)

// evaluate_gate_constraints_base_batch folded with the alphas, on the device; d_out: [nc][local LDE length]
void gate_terms_device(vpbs_ctx* ctx, vpbs_batch* cs, vpbs_batch* wires, const vpbs_gate* gs, unsigned n_gates, unsigned num_selectors,
                       const u64 pi_hash[4], const u64* alphas, unsigned nc, u64* d_out) {
    VPBS_REQUIRE(cs && wires && cs->ctx == ctx && wires->ctx == ctx, "batches of another context");
    VPBS_REQUIRE(cs->log_n == wires->log_n && cs->n_shards == wires->n_shards && cs->shard == wires->shard, "gate terms: batches differ in shape");
    VPBS_REQUIRE(nc >= 1 && nc <= 4, "unsupported number of challenges");
    vpbs::validate_gates(gs, n_gates, num_selectors, cs->ncols, wires->ncols);
    unsigned stride = 1;
    for (unsigned i = 0; i < n_gates; ++i) stride = std::max(stride, gs[i].num_constraints);
    std::vector<u64> h_apow((size_t)nc * stride);
    for (unsigned a = 0; a < nc; ++a) {
        u64 p = 1;
        for (unsigned i = 0; i < stride; ++i) {
            h_apow[(size_t)a * stride + i] = p;
            p = gl::mul(p, alphas[a]);
        }
    }
    u64* d_apow = ctx->alloc_words(h_apow.size());
    // three lanes (streams) when the gate set is large enough to be worth the fork / join (vpbs_ctx_set_gate_lanes; the environment
    // variable VPBS_GATE_LANES=1 makes one stream the default of every context)
    const bool multi = ctx->gate_lanes == 3;
    const size_t len = wires->lde_len();
    u64* lane_buf = nullptr;
    const unsigned n_planes = ctx->tune.gates_fused ? vpbs::gate_terms_planes(gs, n_gates, num_selectors, ctx->tune, len) : 0;
    if (n_planes) {  // one launch (the LDS-tile kernel: one plane)
        u64* planes = nullptr;
        try {
            planes = ctx->alloc_words((size_t)n_planes * nc * len);
            VPBS_HIP(hipMemcpyAsync(d_apow, h_apow.data(), sizeof(u64) * h_apow.size(), hipMemcpyHostToDevice, ctx->stream));
            Timed t(ctx, "gate_constraints");
            vpbs::launch_gate_terms_fused(ctx->stream, ctx->tune, wires->d_lde, cs->d_lde, len, gs, n_gates, num_selectors, pi_hash, d_apow, stride, nc, planes);
            vpbs::launch_sum_planes(ctx->stream, planes, n_planes, (size_t)nc * len, d_out);
            VPBS_HIP(hipGetLastError());
        } catch (...) {
            (void)vpbs::stream_sync(ctx->stream);
            ctx->release(d_apow);
            if (planes) ctx->release(planes);
            throw;
        }
        ctx->release(d_apow);
        ctx->release(planes);
        return;
    }
    try {
        VPBS_HIP(hipMemcpyAsync(d_apow, h_apow.data(), sizeof(u64) * h_apow.size(), hipMemcpyHostToDevice, ctx->stream));
        vpbs::GateLanes lanes{};
        const bool use_lanes = multi && n_gates >= 4;
        if (use_lanes) {
            ctx->ensure_gate_lanes();
            lane_buf = ctx->alloc_words(2 * (size_t)nc * len);
            lanes = {{ctx->gate_streams[0], ctx->gate_streams[1]}, ctx->gate_fork, {ctx->gate_join[0], ctx->gate_join[1]},
                     {lane_buf, lane_buf + (size_t)nc * len}};
        }
        Timed t(ctx, "gate_constraints");
        vpbs::launch_gate_terms(ctx->stream, wires->d_lde, cs->d_lde, len, gs, n_gates, num_selectors, pi_hash, d_apow, stride, nc, d_out,
                                use_lanes ? &lanes : nullptr);
        VPBS_HIP(hipGetLastError());
    } catch (...) {
        (void)vpbs::stream_sync(ctx->stream);
        for (auto st : ctx->gate_streams)
            if (st) (void)vpbs::stream_sync(st);
        ctx->release(d_apow);
        if (lane_buf) ctx->release(lane_buf);
        throw;
    }
    // the helper streams were joined into ctx->stream, so stream-ordered reuse of these blocks is safe
    ctx->release(d_apow);
    if (lane_buf) ctx->release(lane_buf);
}

// the circuit's gates, evaluated CONCURRENTLY with the permutation part of the quotient (three streams, fork / join on ctx->stream)
struct GateWork {
    const vpbs_gate* gates;
    unsigned n_gates, num_selectors;
    const u64* pi_hash;
};
struct QuotientValuesCall {  // launch_quotient_values(raw) as the "extra" item of the gate lanes
    const u64 *wires, *sigmas, *zs_pp, *roots, *l0, *d_apow, *betas, *gammas;
    unsigned n_routed, log_n, rate_bits, max_degree, nc;
    size_t leaf_offset, local_len;
    u64* q_local;
    static void run(hipStream_t s, void* p) {
        auto* c = static_cast<QuotientValuesCall*>(p);
        vpbs::launch_quotient_values(s, c->wires, c->sigmas, c->zs_pp, c->roots, c->l0, nullptr, c->d_apow, c->betas, c->gammas, c->n_routed, c->log_n,
                                     c->rate_bits, c->max_degree, c->nc, c->leaf_offset, c->local_len, c->q_local, true);
    }
};

// compute_quotient_polys on the device; d_out: [nc * 2^rate_bits][n] coefficient chunks.  Gate terms: none, precomputed
// (d_gate_terms), or evaluated here next to the permutation part (gw)
void quotient_permutation_device(vpbs_ctx* ctx, vpbs_batch* cs, unsigned n_constants, vpbs_batch* wires, vpbs_batch* zs_pp,
                                 unsigned n_routed, const u64* betas, const u64* gammas, const u64* alphas, unsigned nc,
                                 unsigned max_degree, const u64* d_gate_terms, u64* d_out, const vpbs_comm* comm = nullptr,
                                 const GateWork* gw = nullptr) {
    VPBS_REQUIRE(cs && wires && zs_pp && cs->ctx == ctx && wires->ctx == ctx && zs_pp->ctx == ctx, "batches of another context");
    const unsigned world = wires->n_shards;
    VPBS_REQUIRE(cs->n_shards == world && zs_pp->n_shards == world && cs->shard == wires->shard && zs_pp->shard == wires->shard,
                 "quotient: the three batches must be sharded the same way");
    if (world > 1)
        VPBS_REQUIRE(comm && comm->world == world && comm->rank == wires->shard && comm->allgather_dev && comm->d_stage_local &&
                         comm->d_stage_full, "quotient over sharded batches needs comm->allgather_dev and its device staging buffers");
    const unsigned log_n = wires->log_n, rate_bits = ctx->rate_bits;
    VPBS_REQUIRE(cs->log_n == log_n && zs_pp->log_n == log_n && log_n >= 1, "degree mismatch");
    VPBS_REQUIRE(nc >= 1 && nc <= 4 && rate_bits <= 3 && max_degree >= 1, "unsupported quotient shape");
    VPBS_REQUIRE(n_routed >= 1 && n_routed <= wires->ncols && n_constants + n_routed <= cs->ncols, "routed wires / sigma columns out of range");
    const unsigned n_chunks = (n_routed + max_degree - 1) / max_degree;
    VPBS_REQUIRE(zs_pp->ncols == nc * n_chunks, "Z/partial-product batch has the wrong number of columns");
    const unsigned log_big = log_n + rate_bits;
    const size_t big = (size_t)1 << log_big;
    const unsigned n_terms = nc + nc * n_chunks;
    hipStream_t s = ctx->stream;
    std::vector<u64> h_apow((size_t)nc * (n_terms + 1));
    for (unsigned a = 0; a < nc; ++a) {
        u64 p = 1;
        for (unsigned i = 0; i <= n_terms; ++i) {
            h_apow[(size_t)a * (n_terms + 1) + i] = p;
            p = gl::mul(p, alphas[a]);
        }
    }
    u64* d_apow = ctx->alloc_words(h_apow.size());
    u64* q_leaf = ctx->alloc_words((size_t)nc * big);
    u64* q_nat = ctx->alloc_words((size_t)nc * big);
    u64 *d_gpow = nullptr, *lane_buf = nullptr;
    const size_t local_len = wires->lde_len(), leaf_offset = wires->leaf_offset();
    try {
        VPBS_HIP(hipMemcpyAsync(d_apow, h_apow.data(), sizeof(u64) * h_apow.size(), hipMemcpyHostToDevice, s));
        const u64* l0 = ctx->l0_table(log_n);
        const u64* gathered = q_leaf;
        u64* q_local = world > 1 ? comm->d_stage_local : q_leaf;
        if (world > 1) VPBS_REQUIRE((size_t)nc * local_len <= comm->stage_capacity_words, "comm staging buffers too small for the quotient values");
        if (gw) {
            // gate constraints on three streams with the permutation part riding along as one more independent item
            vpbs::validate_gates(gw->gates, gw->n_gates, gw->num_selectors, cs->ncols, wires->ncols);
            VPBS_REQUIRE(gw->num_selectors <= n_constants, "selector columns must be leading constants columns");
            unsigned stride = 1;
            for (unsigned i = 0; i < gw->n_gates; ++i) stride = std::max(stride, gw->gates[i].num_constraints);
            std::vector<u64> h_gpow((size_t)nc * stride);
            for (unsigned a = 0; a < nc; ++a) {
                u64 p = 1;
                for (unsigned i = 0; i < stride; ++i) {
                    h_gpow[(size_t)a * stride + i] = p;
                    p = gl::mul(p, alphas[a]);
                }
            }
            d_gpow = ctx->alloc_words(h_gpow.size());
            const unsigned n_planes = ctx->tune.gates_fused ? vpbs::gate_terms_planes(gw->gates, gw->n_gates, gw->num_selectors, ctx->tune, local_len) : 0;
            lane_buf = ctx->alloc_words(std::max(3u, n_planes) * (size_t)nc * local_len);
            VPBS_HIP(hipMemcpyAsync(d_gpow, h_gpow.data(), sizeof(u64) * h_gpow.size(), hipMemcpyHostToDevice, s));
            ctx->ensure_gate_lanes();
            u64 apow_last[4] = {0, 0, 0, 0};
            for (unsigned a = 0; a < nc; ++a) apow_last[a] = h_apow[(size_t)a * (n_terms + 1) + n_terms];
            if (n_planes) {
                // every gate in ONE launch on the context's stream; the permutation part (reads the sigma and Z columns, few
                // instructions) runs beside it on a helper stream; quotient_combine joins the two
                Timed t(ctx, "gate_constraints");
                VPBS_HIP(hipEventRecord(ctx->gate_fork, s));
                VPBS_HIP(hipStreamWaitEvent(ctx->gate_streams[0], ctx->gate_fork, 0));
                vpbs::launch_quotient_values(ctx->gate_streams[0], wires->d_lde, cs->d_lde + (size_t)n_constants * local_len, zs_pp->d_lde,
                                             ctx->roots(log_big, false), l0, nullptr, d_apow, betas, gammas, n_routed, log_n, rate_bits, max_degree, nc,
                                             leaf_offset, local_len, q_local, true);
                VPBS_HIP(hipEventRecord(ctx->gate_join[0], ctx->gate_streams[0]));
                vpbs::launch_gate_terms_fused(s, ctx->tune, wires->d_lde, cs->d_lde, local_len, gw->gates, gw->n_gates, gw->num_selectors, gw->pi_hash, d_gpow, stride,
                                              nc, lane_buf);
                VPBS_HIP(hipStreamWaitEvent(s, ctx->gate_join[0], 0));
                vpbs::launch_quotient_combine_planes(s, q_local, lane_buf, n_planes, apow_last, log_n, rate_bits, nc, leaf_offset, local_len);
            } else {
            QuotientValuesCall call{wires->d_lde, cs->d_lde + (size_t)n_constants * local_len, zs_pp->d_lde, ctx->roots(log_big, false), l0, d_apow,
                                    betas, gammas, n_routed, log_n, rate_bits, max_degree, nc, leaf_offset, local_len, q_local};
            vpbs::GateLanes lanes{};
            lanes.stream[0] = ctx->gate_streams[0];
            lanes.stream[1] = ctx->gate_streams[1];
            lanes.fork = ctx->gate_fork;
            lanes.join[0] = ctx->gate_join[0];
            lanes.join[1] = ctx->gate_join[1];
            lanes.out[0] = lane_buf + (size_t)nc * local_len;
            lanes.out[1] = lane_buf + 2 * (size_t)nc * local_len;
            lanes.extra = &QuotientValuesCall::run;
            lanes.extra_arg = &call;
            lanes.extra_weight = 165;
            lanes.skip_sum = true;
            Timed t(ctx, "gate_constraints");  // gates + permutation part, overlapped
            vpbs::launch_gate_terms(s, wires->d_lde, cs->d_lde, local_len, gw->gates, gw->n_gates, gw->num_selectors, gw->pi_hash, d_gpow, stride, nc,
                                    lane_buf, &lanes);
            vpbs::launch_quotient_combine(s, q_local, lanes.used[0] ? lane_buf : nullptr, lanes.used[1] ? lanes.out[0] : nullptr,
                                          lanes.used[2] ? lanes.out[1] : nullptr, apow_last, log_n, rate_bits, nc, leaf_offset, local_len);
            }
        } else {
            Timed t(ctx, "quotient_permutation");
            vpbs::launch_quotient_values(s, wires->d_lde, cs->d_lde + (size_t)n_constants * local_len, zs_pp->d_lde, ctx->roots(log_big, false), l0,
                                         d_gate_terms, d_apow, betas, gammas, n_routed, log_n, rate_bits, max_degree, nc, leaf_offset, local_len,
                                         q_local);
        }
        if (world > 1) {
            // every rank needs all 8n values for the (cheap, replicated) inverse transform: device all-gather, 16 B per point
            VPBS_HIP(vpbs::stream_sync(s));
            if (comm->allgather_dev(comm->user, (size_t)nc * local_len) != 0) throw DeviceError{VPBS_ERR_DEVICE, "quotient all-gather failed"};
            gathered = comm->d_stage_full;
        }
        {
            Timed t(ctx, "quotient_permutation");
            vpbs::launch_quotient_finish(s, gathered, local_len, ctx->roots(log_big, true), ctx->prescale(log_big, 0, gl::inv(gl::GENERATOR)), log_n,
                                         rate_bits, nc, q_nat, q_leaf, d_out);
        }
        VPBS_HIP(hipGetLastError());
        if (world > 1) VPBS_HIP(vpbs::stream_sync(s));  // the staging buffers belong to the communicator: done with them
    } catch (...) {
        (void)vpbs::stream_sync(s);
        for (auto st : ctx->gate_streams)
            if (st) (void)vpbs::stream_sync(st);
        ctx->release(d_apow); ctx->release(q_leaf); ctx->release(q_nat);
        if (d_gpow) ctx->release(d_gpow);
        if (lane_buf) ctx->release(lane_buf);
        throw;
    }
    // stream-ordered reuse of the scratch blocks is safe: later work is enqueued on the same stream (the helper streams were joined)
    ctx->release(d_apow);
    ctx->release(q_leaf);
    ctx->release(q_nat);
    if (d_gpow) ctx->release(d_gpow);
    if (lane_buf) ctx->release(lane_buf);
}
}  // namespace

// ---------------- C ABI ----------------
template <typename F>
static int guarded(vpbs_ctx* ctx, F&& f) {
    try {
        if (ctx) VPBS_HIP(hipSetDevice(ctx->device));  // the current device is per host thread
        f();
        return VPBS_OK;
    } catch (const DeviceError& e) {
        if (ctx) ctx->err = e.what;
        return e.status;
    } catch (const std::exception& e) {
        if (ctx) ctx->err = e.what();
        return VPBS_ERR_INVALID;
    }
}

static plonky2::FriInstanceInfo instance_from_c(const vpbs_fri_instance* in) {
    plonky2::FriInstanceInfo inst;
    for (size_t b = 0; b < in->n_batches; ++b) {
        plonky2::FriBatchInfo bi;
        bi.point = gl::Ext{in->batches[b].point[0], in->batches[b].point[1]};
        for (size_t j = 0; j < in->batches[b].n_polys; ++j)
            bi.polynomials.push_back({in->batches[b].oracle_index[j], in->batches[b].poly_index[j]});
        inst.batches.push_back(std::move(bi));
    }
    return inst;
}

// plonk/circuit_data.rs get_fri_instance: batch 0 = every polynomial of every oracle at zeta;
// batch 1 = the num_challenges Z polynomials (oracle 2, first columns) at g * zeta
static plonky2::FriInstanceInfo step_fri_instance(const std::vector<size_t>& ncols, unsigned num_challenges, gl::Ext zeta, unsigned degree_bits) {
    plonky2::FriInstanceInfo inst;
    plonky2::FriBatchInfo all, next;
    all.point = zeta;
    for (u32 o = 0; o < ncols.size(); ++o)
        for (u32 p = 0; p < ncols[o]; ++p) all.polynomials.push_back({o, p});
    // F::Extension::primitive_root_of_unity(degree_bits) lies in the base field (EXT generator squared = base generator)
    next.point = gl::mul(zeta, gl::root_of_unity(degree_bits));
    for (u32 p = 0; p < num_challenges; ++p) next.polynomials.push_back({2, p});
    inst.batches = {all, next};
    return inst;
}

extern "C" {

void vpbs_fri_params_standard(unsigned degree_bits, vpbs_fri_params* out) {
    const plonky2::FriParams p = plonky2::FriParams::standard(degree_bits);
    std::memset(out, 0, sizeof *out);
    out->rate_bits = p.config.rate_bits;
    out->cap_height = p.config.cap_height;
    out->pow_bits = p.config.proof_of_work_bits;
    out->num_query_rounds = p.config.num_query_rounds;
    out->n_rounds = (unsigned)p.reduction_arity_bits.size();
    for (unsigned i = 0; i < out->n_rounds; ++i) out->arity_bits[i] = p.reduction_arity_bits[i];
}

size_t vpbs_fri_proof_words(const vpbs_fri_params* params, unsigned degree_bits, const size_t* ncols, size_t n_oracles) {
    return plonky2::fri_proof_words(plonky2::FriParams::from_c(*params, degree_bits), std::vector<size_t>(ncols, ncols + n_oracles));
}

int vpbs_fri_prove(vpbs_ctx* ctx, vpbs_batch* const* oracles, size_t n_oracles, const vpbs_fri_instance* instance,
                   const vpbs_fri_params* params, vpbs_challenger_state* challenger, uint64_t forced_pow, uint64_t* proof_out) {
    if (!ctx || !oracles || !n_oracles || !instance || !params || !challenger || !proof_out) return VPBS_ERR_INVALID;
    return guarded(ctx, [&] {
        VPBS_HIP(hipSetDevice(ctx->device));
        plonky2::Challenger ch(*challenger);
        std::vector<vpbs_batch*> os(oracles, oracles + n_oracles);
        plonky2::PolynomialBatch::prove_openings(ctx, instance_from_c(instance), os, ch,
                                                 plonky2::FriParams::from_c(*params, os[0]->log_n), forced_pow, proof_out);
        *challenger = ch.st;
    });
}

int vpbs_partial_products(vpbs_ctx* ctx, const uint64_t* wires, const uint64_t* sigmas, int on_device, unsigned n_routed, unsigned log_n,
                          const uint64_t* betas, const uint64_t* gammas, unsigned num_challenges, unsigned max_degree, uint64_t* out) {
    if (!ctx || !wires || !sigmas || !betas || !gammas || !out || n_routed == 0 || max_degree == 0 || num_challenges == 0 || log_n == 0)
        return VPBS_ERR_INVALID;
    return guarded(ctx, [&] {
        const size_t n = (size_t)1 << log_n;
        const size_t chunks = (n_routed + max_degree - 1) / max_degree;
        const size_t out_words = (size_t)num_challenges * chunks * n;
        std::vector<void*> tmp;
        struct Cleanup {
            vpbs_ctx* c;
            std::vector<void*>& v;
            ~Cleanup() {
                (void)vpbs::stream_sync(c->stream);
                for (void* p : v) c->release(p);
            }
        } cleanup{ctx, tmp};
        const u64 *d_w = wires, *d_s = sigmas;
        u64* d_o = out;
        if (!on_device) {
            u64* a = ctx->alloc_words((size_t)n_routed * n);
            tmp.push_back(a);
            u64* b = ctx->alloc_words((size_t)n_routed * n);
            tmp.push_back(b);
            d_o = ctx->alloc_words(out_words);
            tmp.push_back(d_o);
            VPBS_HIP(hipMemcpyAsync(a, wires, sizeof(u64) * n_routed * n, hipMemcpyHostToDevice, ctx->stream));
            VPBS_HIP(hipMemcpyAsync(b, sigmas, sizeof(u64) * n_routed * n, hipMemcpyHostToDevice, ctx->stream));
            d_w = a;
            d_s = b;
        }
        partial_products_device(ctx, d_w, d_s, n_routed, log_n, betas, gammas, num_challenges, max_degree, d_o);
        if (!on_device) {
            VPBS_HIP(hipMemcpyAsync(out, d_o, sizeof(u64) * out_words, hipMemcpyDeviceToHost, ctx->stream));
            VPBS_HIP(vpbs::stream_sync(ctx->stream));
        }
    });
}

int vpbs_quotient_permutation(vpbs_ctx* ctx, vpbs_batch* cs, unsigned n_constants, vpbs_batch* wires, vpbs_batch* zs_pp, unsigned n_routed,
                              const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas, unsigned nc, unsigned max_degree,
                              const uint64_t* d_gate_terms, uint64_t* out, int out_on_device) {
    if (!ctx || !cs || !wires || !zs_pp || !betas || !gammas || !alphas || !out) return VPBS_ERR_INVALID;
    return guarded(ctx, [&] {
        const size_t words = ((size_t)nc << ctx->rate_bits) << wires->log_n;
        if (out_on_device) {
            quotient_permutation_device(ctx, cs, n_constants, wires, zs_pp, n_routed, betas, gammas, alphas, nc, max_degree, d_gate_terms, out);
            return;
        }
        u64* d_out = ctx->alloc_words(words);
        try {
            quotient_permutation_device(ctx, cs, n_constants, wires, zs_pp, n_routed, betas, gammas, alphas, nc, max_degree, d_gate_terms, d_out);
            VPBS_HIP(hipMemcpyAsync(out, d_out, sizeof(u64) * words, hipMemcpyDeviceToHost, ctx->stream));
            VPBS_HIP(vpbs::stream_sync(ctx->stream));
        } catch (...) {
            (void)vpbs::stream_sync(ctx->stream);
            ctx->release(d_out);
            throw;
        }
        ctx->release(d_out);
    });
}

int vpbs_gate_terms(vpbs_ctx* ctx, vpbs_batch* cs, vpbs_batch* wires, const vpbs_gate* gates, unsigned n_gates, unsigned num_selectors,
                    const uint64_t pi_hash[4], const uint64_t* alphas, unsigned nc, uint64_t* d_out) {
    if (!ctx || !cs || !wires || !gates || !pi_hash || !alphas || !d_out) return VPBS_ERR_INVALID;
    return guarded(ctx, [&] { gate_terms_device(ctx, cs, wires, gates, n_gates, num_selectors, pi_hash, alphas, nc, d_out); });
}

int vpbs_step_sizes_get(const vpbs_ctx* ctx, const vpbs_step_inputs* in, vpbs_step_sizes* out) {
    if (!ctx || !in || !out || !in->constants_sigmas) return VPBS_ERR_INVALID;
    const plonky2::FriParams fp = plonky2::FriParams::standard(in->log_n, ctx->rate_bits, ctx->cap_height);
    const size_t n_cs = in->constants_sigmas->ncols;
    out->cap_words = (size_t)4 << ctx->cap_height;
    out->openings_words = 2 * (n_cs + in->n_wires + in->n_zs_partial_products + in->n_quotient + in->num_challenges);
    out->fri_words = plonky2::fri_proof_words(fp, {n_cs, in->n_wires, in->n_zs_partial_products, in->n_quotient});
    return VPBS_OK;
}

namespace {
// The collectives of ONE sharded step proof, with a status word per rank riding on every host collective (include/vpbs_prover.h, "failure
// semantics").  `outer` is the communicator the prover's stages see: its callbacks append this rank's status to the payload, strip the
// peers' on the way back and note a failure; `plan` is the fixed sequence of collectives of the step, so that a rank that fails (or learns
// that a peer has) can take part in what is left of it with zeros (finish()), instead of leaving the others waiting.
struct ShardSession {
    enum Kind { GATHER, GATHER_DEV, REDUCE };
    struct Coll {
        Kind kind;
        size_t words;
    };
    const vpbs_comm* inner;
    vpbs_comm outer;
    std::vector<Coll> plan;
    size_t next = 0;
    int status = 0;           // this rank's: 0 while it is fine
    bool peer_failed = false, broken = false;   // broken: a collective itself failed -- nothing more can be exchanged
    std::vector<u64> buf_a, buf_b;

    ShardSession(const vpbs_comm* c, size_t cap_local_words, size_t quotient_dev_words, size_t query_words) : inner(c), outer(*c) {
        plan = {{GATHER, cap_local_words}, {GATHER, cap_local_words}};
        if (quotient_dev_words) plan.push_back({GATHER_DEV, quotient_dev_words});
        plan.push_back({GATHER, cap_local_words});
        plan.push_back({REDUCE, query_words});
        outer.user = this;
        outer.allgather = [](void* u, const uint64_t* local, size_t words, uint64_t* full) { return static_cast<ShardSession*>(u)->gather(local, words, full); };
        outer.allreduce_sum = [](void* u, uint64_t* inout, size_t words) { return static_cast<ShardSession*>(u)->reduce(inout, words); };
        if (c->allgather_dev) outer.allgather_dev = [](void* u, size_t words) { return static_cast<ShardSession*>(u)->gather_dev(words); };
    }
    int gather(const uint64_t* local, size_t words, uint64_t* full) {
        if (broken || next >= plan.size() || plan[next].kind != GATHER || plan[next].words != words) return broken = true, -1;
        ++next;
        const unsigned world = inner->world;
        buf_a.assign(words + 1, 0);
        if (local) std::memcpy(buf_a.data(), local, 8 * words);
        buf_a[words] = status ? 1 : 0;
        buf_b.assign((size_t)world * (words + 1), 0);
        if (inner->allgather(inner->user, buf_a.data(), words + 1, buf_b.data()) != 0) return broken = true, -1;
        for (unsigned r = 0; r < world; ++r) {
            if (full) std::memcpy(full + (size_t)r * words, buf_b.data() + (size_t)r * (words + 1), 8 * words);
            if (r != inner->rank && buf_b[(size_t)r * (words + 1) + words] != 0) peer_failed = true;
        }
        return peer_failed ? -1 : 0;
    }
    int reduce(uint64_t* inout, size_t words) {
        if (broken || next >= plan.size() || plan[next].kind != REDUCE || plan[next].words != words) return broken = true, -1;
        ++next;
        buf_a.assign(words + 1, 0);
        if (inout) std::memcpy(buf_a.data(), inout, 8 * words);
        buf_a[words] = status ? 1 : 0;
        if (inner->allreduce_sum(inner->user, buf_a.data(), words + 1) != 0) return broken = true, -1;
        if (inout) std::memcpy(inout, buf_a.data(), 8 * words);
        if (buf_a[words] != (status ? 1u : 0u)) peer_failed = true;
        return peer_failed ? -1 : 0;
    }
    int gather_dev(size_t words) {   // device payload: no status of its own (a failure shows at the next host collective)
        if (broken || next >= plan.size() || plan[next].kind != GATHER_DEV || plan[next].words != words) return broken = true, -1;
        ++next;
        if (inner->allgather_dev(inner->user, words) != 0) return broken = true, -1;
        return 0;
    }
    // this rank is out (its own failure, or a peer's): the collectives the step has left, with zeros and the status
    void finish(int why) {
        if (!status) status = why ? why : VPBS_ERR_PEER;
        while (!broken && next < plan.size()) {
            const Coll c = plan[next];
            if (c.kind == GATHER) (void)gather(nullptr, c.words, nullptr);
            else if (c.kind == REDUCE) (void)reduce(nullptr, c.words);
            else if (inner->allgather_dev) (void)gather_dev(c.words);
            else broken = true;
        }
    }
};

// shape of the step's collectives from its inputs: words of a rank's cap share, of its quotient values (0: quotient supplied), of the query records
void shard_plan_sizes(const vpbs_ctx* ctx, const vpbs_step_inputs* in, unsigned world, size_t& cap_local, size_t& quot_dev, size_t& query_words) {
    const plonky2::FriParams fp = plonky2::FriParams::standard(in->log_n, ctx->rate_bits, ctx->cap_height);
    const size_t cap_words = (size_t)4 << ctx->cap_height, n_cs = in->constants_sigmas->ncols;
    cap_local = cap_words / world;
    quot_dev = in->quotient_coeffs ? 0 : (size_t)in->num_challenges * (((size_t)1 << (in->log_n + ctx->rate_bits)) / world);
    const size_t total = plonky2::fri_proof_words(fp, {n_cs, in->n_wires, in->n_zs_partial_products, in->n_quotient});
    query_words = total - fp.reduction_arity_bits.size() * cap_words - ((size_t)2 << fp.final_poly_bits()) - 1;
}

// fault injection for the failure-semantics tests (the reference has no failure handling to mirror: SURVEY.md section 5): VPBS_FAULT_INJECT=
// "<rank>:<stage>" makes that rank of a sharded step throw before its <stage>-th commitment (1 = wires, 2 = Z / partial products, 3 = quotient)
void maybe_inject_fault(const vpbs_comm* comm, int stage) {
    if (!comm) return;
    const char* const e = std::getenv("VPBS_FAULT_INJECT");   // read at every sharded step: a test switches it between two steps
    if (!e) return;
    unsigned rank = 0;
    int at = 0;
    if (std::sscanf(e, "%u:%d", &rank, &at) == 2 && rank == comm->rank && at == stage)
        throw DeviceError{VPBS_ERR_DEVICE, "injected failure before commitment " + std::to_string(stage) + " (VPBS_FAULT_INJECT)"};
}
}  // namespace

static int prove_step_impl(vpbs_ctx* ctx, const vpbs_step_inputs* in, const vpbs_comm* comm, uint64_t* caps_out, uint64_t* openings_out,
                           uint64_t* fri_out, vpbs_challenger_state* challenger_out, uint64_t* challenges_out);

// the sharded form: the same prover under a ShardSession, which sees to it that every rank leaves the step through all of its collectives
static int prove_step_sharded_impl(vpbs_ctx* ctx, const vpbs_step_inputs* in, const vpbs_comm* comm, uint64_t* caps_out, uint64_t* openings_out,
                                   uint64_t* fri_out, vpbs_challenger_state* challenger_out, uint64_t* challenges_out) {
    if (!ctx || !in || !in->constants_sigmas || !comm || comm->world < 2 || comm->rank >= comm->world || !comm->allgather || !comm->allreduce_sum ||
        (((size_t)4 << ctx->cap_height) % comm->world) != 0)
        return VPBS_ERR_INVALID;
    size_t cap_local, quot_dev, query_words;
    shard_plan_sizes(ctx, in, comm->world, cap_local, quot_dev, query_words);
    ShardSession session(comm, cap_local, quot_dev, query_words);
    int rc = prove_step_impl(ctx, in, &session.outer, caps_out, openings_out, fri_out, challenger_out, challenges_out);
    if (rc == VPBS_OK && session.next != session.plan.size()) {
        ctx->err = "sharded step: the prover left collectives of the step undone";   // cannot happen with matching shapes; do not leave peers waiting
        rc = VPBS_ERR_INVALID;
    }
    if (rc != VPBS_OK) {
        const bool peer = session.peer_failed && !session.status;
        session.finish(peer ? VPBS_ERR_PEER : rc);
        if (peer) {
            ctx->err = "sharded step: another rank failed (" + ctx->err + ")";
            rc = VPBS_ERR_PEER;
        } else if (session.broken) {
            ctx->err += " [a collective itself failed: peers may be waiting until their communicator's timeout]";
        }
    }
    return rc;
}

static int prove_step_impl(vpbs_ctx* ctx, const vpbs_step_inputs* in, const vpbs_comm* comm, uint64_t* caps_out, uint64_t* openings_out,
                           uint64_t* fri_out, vpbs_challenger_state* challenger_out, uint64_t* challenges_out) {
    if (!ctx || !in || !caps_out || !openings_out || !fri_out || !in->constants_sigmas) return VPBS_ERR_INVALID;
    if (comm && (comm->world == 0 || comm->rank >= comm->world || (comm->world > 1 && (!comm->allgather || !comm->allreduce_sum))))
        return VPBS_ERR_INVALID;
    if (comm && comm->world == 1) comm = nullptr;
    return guarded(ctx, [&] {
        using namespace plonky2;
        VPBS_HIP(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        const unsigned log_n = in->log_n, nc = in->num_challenges;
        const size_t n = (size_t)1 << log_n;
        VPBS_REQUIRE(in->constants_sigmas->log_n == log_n, "constants_sigmas degree mismatch");
        VPBS_REQUIRE(nc >= 1 && nc <= 4, "num_challenges must be 1 .. 4");
        VPBS_REQUIRE(nc <= in->n_zs_partial_products, "num_challenges exceeds the Z/partial-product batch");
        const size_t cap_words = (size_t)4 << ctx->cap_height;

        // stage the three matrices on the device if they arrive from the host
        std::vector<void*> staged;
        auto on_device = [&](const u64* p, size_t words) -> const u64* {
            if (in->inputs_on_device) return p;
            u64* d = ctx->alloc_words(words);
            staged.push_back(d);
            VPBS_HIP(hipMemcpyAsync(d, p, sizeof(u64) * words, hipMemcpyHostToDevice, s));
            return d;
        };
        struct Cleanup {
            vpbs_ctx* c;
            std::vector<void*>& v;
            ~Cleanup() {
                (void)vpbs::stream_sync(c->stream);
                for (void* p : v) c->release(p);
            }
        } cleanup{ctx, staged};
        const u64* d_wires = on_device(in->wires_values, (size_t)in->n_wires * n);
        const u64* d_zs = in->zs_pp_values ? on_device(in->zs_pp_values, (size_t)in->n_zs_partial_products * n) : nullptr;
        const u64* d_quot = in->quotient_coeffs ? on_device(in->quotient_coeffs, (size_t)in->n_quotient * n) : nullptr;
        if (!in->quotient_coeffs) {
            VPBS_REQUIRE(in->n_quotient == nc * (1u << ctx->rate_bits) && in->n_routed >= 1 && in->quotient_degree_factor >= 1,
                         "quotient_coeffs == NULL needs n_quotient = num_challenges * 8, n_routed and quotient_degree_factor");
        }
        const u64* d_sigmas = nullptr;
        if (!in->zs_pp_values) {
            VPBS_REQUIRE(in->sigmas_values && in->n_routed >= 1 && in->n_routed <= in->n_wires && in->quotient_degree_factor >= 1,
                         "zs_pp_values == NULL needs sigmas_values, n_routed and quotient_degree_factor");
            const unsigned chunks = (in->n_routed + in->quotient_degree_factor - 1) / in->quotient_degree_factor;
            VPBS_REQUIRE(in->n_zs_partial_products == nc * chunks, "n_zs_partial_products != num_challenges * (num_partial_products + 1)");
            d_sigmas = in->sigmas_on_device ? in->sigmas_values : on_device(in->sigmas_values, (size_t)in->n_routed * n);
        }

        // prove(): public_inputs_hash, wires commitment, transcript
        maybe_inject_fault(comm, 1);
        PolynomialBatch wires = PolynomialBatch::from_values(ctx, d_wires, in->n_wires, log_n, false, comm);
        HashOut pi_hash;  // hashed on the host while the device works on the wires commitment
        vpbs_hash_no_pad(in->public_inputs, in->n_public_inputs, pi_hash.data());
        wires.merkle_cap(caps_out, comm);
        Challenger challenger;
        challenger.observe_elements(in->circuit_digest, 4);
        challenger.observe_hash(pi_hash);
        challenger.observe_cap(caps_out, cap_words / 4);
        const std::vector<u64> betas = challenger.get_n_challenges(nc);
        const std::vector<u64> gammas = challenger.get_n_challenges(nc);
        // all_wires_permutation_partial_products(betas, gammas): on the device when no precomputed matrix is supplied
        volatile unsigned* pp_flag = nullptr;
        if (!d_zs) {
            u64* d_pp = ctx->alloc_words((size_t)in->n_zs_partial_products * n);
            staged.push_back(d_pp);
            partial_products_device(ctx, d_wires, d_sigmas, in->n_routed, log_n, betas.data(), gammas.data(), nc,
                                    in->quotient_degree_factor, d_pp, &pp_flag);
            d_zs = d_pp;
        }
        maybe_inject_fault(comm, 2);
        PolynomialBatch zs_pp = PolynomialBatch::from_values(ctx, d_zs, in->n_zs_partial_products, log_n, false, comm);
        zs_pp.merkle_cap(caps_out + cap_words, comm);
        // the flag of the partial products arrived with the cap (one round trip instead of two)
        VPBS_REQUIRE(!pp_flag || *pp_flag == 0, "zero denominator in the permutation argument (the reference's batch inverse would panic)");
        challenger.observe_cap(caps_out + cap_words, cap_words / 4);
        const std::vector<u64> alphas = challenger.get_n_challenges(nc);
        // compute_quotient_polys(alphas): supplied coefficient chunks, or evaluated on the device (permutation argument +
        // the gate constraints of in->gates)
        if (!d_quot) {
            u64* d_q = ctx->alloc_words((size_t)in->n_quotient * n);
            staged.push_back(d_q);
            u64* d_gate = nullptr;
            const bool overlapped = in->gates && in->n_gates >= 4 && ctx->gate_lanes == 3;
            if (in->gates && in->n_gates && !overlapped) {  // evaluate_gate_constraints_base_batch for the circuit's gate set
                VPBS_REQUIRE(in->num_selectors <= in->n_constants, "selector columns must be leading constants columns");
                d_gate = ctx->alloc_words((size_t)nc * wires.h->lde_len());
                staged.push_back(d_gate);
                gate_terms_device(ctx, in->constants_sigmas, wires.h, in->gates, in->n_gates, in->num_selectors, pi_hash.data(), alphas.data(), nc, d_gate);
            }
            const GateWork gw{in->gates, in->n_gates, in->num_selectors, pi_hash.data()};
            quotient_permutation_device(ctx, in->constants_sigmas, in->n_constants, wires.h, zs_pp.h, in->n_routed, betas.data(),
                                        gammas.data(), alphas.data(), nc, in->quotient_degree_factor, d_gate, d_q, comm, overlapped ? &gw : nullptr);
            d_quot = d_q;
        }
        maybe_inject_fault(comm, 3);
        PolynomialBatch quotient = PolynomialBatch::from_coeffs(ctx, d_quot, in->n_quotient, log_n, false, comm);
        quotient.merkle_cap(caps_out + 2 * cap_words, comm);
        challenger.observe_cap(caps_out + 2 * cap_words, cap_words / 4);
        const Ext zeta = challenger.get_extension_challenge();
        // ensure!(zeta^(2^degree_bits) != 1, "Opening point is in the subgroup.")
        {
            Ext t = zeta;
            for (unsigned i = 0; i < log_n; ++i) t = gl::mul(t, t);
            VPBS_REQUIRE(!gl::eq(t, gl::ext(1)), "Opening point is in the subgroup.");
        }
        if (challenges_out) {
            u64* c = challenges_out;
            for (u64 v : betas) *c++ = v;
            for (u64 v : gammas) *c++ = v;
            for (u64 v : alphas) *c++ = v;
            *c++ = zeta.c0;
            *c++ = zeta.c1;
        }

        // OpeningSet::new: every polynomial at zeta, the Z polynomials also at g * zeta
        std::vector<vpbs_batch*> oracles = {in->constants_sigmas, wires.h, zs_pp.h, quotient.h};
        std::vector<size_t> ncols;
        size_t total_cols = 0;
        for (auto* o : oracles) {
            ncols.push_back(o->ncols);
            total_cols += o->ncols;
        }
        const FriInstanceInfo instance = step_fri_instance(ncols, nc, zeta, log_n);
        const Ext zeta_next = instance.batches[1].point;
        const unsigned chunks = (unsigned)((n + 4095) / 4096);
        u64* zpow = ctx->alloc_words(4 * n);
        staged.push_back(zpow);   // on the cleanup list before the next allocation can throw
        u64* znpow = zpow + 2 * n;
        u64* d_open = ctx->alloc_words(2 * (total_cols + nc) * (size_t)(1 + chunks) + 64);
        staged.push_back(d_open);
        {
            Timed t(ctx, "openings_eval");
            const Ext pts[2] = {zeta, zeta_next};
            vpbs::launch_ext_powers(s, pts, 2, n, zpow);
            // every oracle at zeta and the Z columns at g * zeta in one pair of launches; results land in openings order
            vpbs::EvalSegments segs{};
            for (auto* o : oracles) segs.seg[segs.count++] = {o->d_coeffs, zpow, n, o->ncols};
            segs.seg[segs.count++] = {zs_pp.h->d_coeffs, znpow, n, nc};
            vpbs::launch_eval_ext_multi(s, segs, n, d_open);
        }
        ctx->d2h_sync(openings_out, d_open, sizeof(u64) * 2 * (total_cols + nc));
        if (in->on_section) in->on_section(in->on_section_user, 1);   // caps and openings are final
        // challenger.observe_openings(&openings.to_fri_openings()): zeta batch then zeta_next batch
        challenger.observe_elements(openings_out, 2 * (total_cols + nc));

        FriParams fp = FriParams::standard(log_n, ctx->rate_bits, ctx->cap_height);
        fp.mul_final_by_x = ctx->compat.fri_mul_final_by_x != 0;   // the switch table of include/vpbs_prover.h
        PolynomialBatch::prove_openings(ctx, instance, oracles, challenger, fp, in->forced_pow, fri_out, comm, in->on_section, in->on_section_user);
        if (challenger_out) *challenger_out = challenger.st;
    });
}

int vpbs_prove_step(vpbs_ctx* ctx, const vpbs_step_inputs* in, uint64_t* caps_out, uint64_t* openings_out, uint64_t* fri_out,
                    vpbs_challenger_state* challenger_out, uint64_t* challenges_out) {
    return prove_step_impl(ctx, in, nullptr, caps_out, openings_out, fri_out, challenger_out, challenges_out);
}
int vpbs_prove_step_sharded(vpbs_ctx* ctx, const vpbs_step_inputs* in, const vpbs_comm* comm, uint64_t* caps_out, uint64_t* openings_out,
                            uint64_t* fri_out, vpbs_challenger_state* challenger_out, uint64_t* challenges_out) {
    if (!comm) return VPBS_ERR_INVALID;
    if (comm->world == 1) return prove_step_impl(ctx, in, nullptr, caps_out, openings_out, fri_out, challenger_out, challenges_out);
    return prove_step_sharded_impl(ctx, in, comm, caps_out, openings_out, fri_out, challenger_out, challenges_out);
}

int vpbs_prove_step_sharded_fail(vpbs_ctx* ctx, const vpbs_step_inputs* in, const vpbs_comm* comm, int status) {
    if (!ctx || !in || !in->constants_sigmas || !comm || comm->world == 0 || comm->rank >= comm->world) return VPBS_ERR_INVALID;
    if (comm->world == 1) return VPBS_OK;
    if (!comm->allgather || !comm->allreduce_sum) return VPBS_ERR_INVALID;
    size_t cap_local, quot_dev, query_words;
    shard_plan_sizes(ctx, in, comm->world, cap_local, quot_dev, query_words);
    ShardSession session(comm, cap_local, quot_dev, query_words);
    session.finish(status ? status : VPBS_ERR_INVALID);
    return session.broken ? VPBS_ERR_DEVICE : VPBS_OK;
}

int vpbs_comm_allgather_checked(const vpbs_comm* comm, const uint64_t* local, size_t local_words, uint64_t* full, int local_status) {
    if (!comm || !comm->allgather || comm->world == 0 || !full || (!local && local_words)) return VPBS_ERR_INVALID;
    std::vector<u64> a(local_words + 1, 0), b((size_t)comm->world * (local_words + 1), 0);
    if (local && !local_status) std::memcpy(a.data(), local, 8 * local_words);
    a[local_words] = local_status ? 1 : 0;
    if (comm->allgather(comm->user, a.data(), local_words + 1, b.data()) != 0) return VPBS_ERR_DEVICE;
    bool peer = false;
    for (unsigned r = 0; r < comm->world; ++r) {
        std::memcpy(full + (size_t)r * local_words, b.data() + (size_t)r * (local_words + 1), 8 * local_words);
        if (r != comm->rank && b[(size_t)r * (local_words + 1) + local_words]) peer = true;
    }
    return local_status ? local_status : (peer ? VPBS_ERR_PEER : VPBS_OK);
}

// ProofWithPublicInputs::to_bytes (util/serialization): caps, OpeningSet, FriProof, then public inputs
long vpbs_step_proof_to_bytes(const vpbs_ctx* ctx, const vpbs_step_inputs* in, unsigned n_constants, const uint64_t* caps,
                              const uint64_t* openings, const uint64_t* fri, uint8_t* out, size_t cap_bytes) {
    if (!ctx || !in || !caps || !openings || !fri || !out || !in->constants_sigmas) return VPBS_ERR_INVALID;
    const plonky2::FriParams fp = plonky2::FriParams::standard(in->log_n, ctx->rate_bits, ctx->cap_height);
    const size_t n_cs = in->constants_sigmas->ncols;
    if (n_constants > n_cs) return VPBS_ERR_INVALID;
    size_t pos = 0;
    bool overflow = false;
    auto put_words = [&](const uint64_t* w, size_t cnt) {
        if (pos + 8 * cnt > cap_bytes) { overflow = true; return; }
        std::memcpy(out + pos, w, 8 * cnt);  // little-endian host
        pos += 8 * cnt;
    };
    auto put_u8 = [&](uint8_t b) {
        if (pos + 1 > cap_bytes) { overflow = true; return; }
        out[pos++] = b;
    };
    const size_t cap_words = (size_t)4 << ctx->cap_height;
    put_words(caps, 3 * cap_words);  // wires_cap, plonk_zs_partial_products_cap, quotient_polys_cap
    // OpeningSet field order: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys,
    // lookup_zs (empty), lookup_zs_next (empty).  `openings` holds [cs | wires | zs_pp | quotient | zs_next].
    const unsigned nc = in->num_challenges;
    const uint64_t* cs = openings;
    const uint64_t* wires = cs + 2 * n_cs;
    const uint64_t* zs_pp = wires + 2 * (size_t)in->n_wires;
    const uint64_t* quot = zs_pp + 2 * (size_t)in->n_zs_partial_products;
    const uint64_t* zs_next = quot + 2 * (size_t)in->n_quotient;
    put_words(cs, 2 * (size_t)n_constants);
    put_words(cs + 2 * (size_t)n_constants, 2 * (n_cs - n_constants));
    put_words(wires, 2 * (size_t)in->n_wires);
    put_words(zs_pp, 2 * (size_t)nc);
    put_words(zs_next, 2 * (size_t)nc);
    put_words(zs_pp + 2 * (size_t)nc, 2 * (size_t)(in->n_zs_partial_products - nc));
    put_words(quot, 2 * (size_t)in->n_quotient);
    // FriProof
    const uint64_t* w = fri;
    const size_t n_rounds = fp.reduction_arity_bits.size();
    put_words(w, n_rounds * cap_words);
    w += n_rounds * cap_words;
    const unsigned log_lde = fp.lde_bits();
    const size_t oracle_cols[4] = {n_cs, in->n_wires, in->n_zs_partial_products, in->n_quotient};
    for (unsigned q = 0; q < fp.config.num_query_rounds; ++q) {
        for (size_t o = 0; o < 4; ++o) {
            const unsigned nsib = log_lde - fp.config.cap_height;
            put_words(w, oracle_cols[o]);
            w += oracle_cols[o];
            put_u8((uint8_t)nsib);
            put_words(w, 4 * (size_t)nsib);
            w += 4 * (size_t)nsib;
        }
        unsigned lg = log_lde;
        for (unsigned ab : fp.reduction_arity_bits) {
            lg -= ab;
            const unsigned nsib = lg - fp.config.cap_height;
            put_words(w, (size_t)2 << ab);
            w += (size_t)2 << ab;
            put_u8((uint8_t)nsib);
            put_words(w, 4 * (size_t)nsib);
            w += 4 * (size_t)nsib;
        }
    }
    const size_t final_words = (size_t)2 << fp.final_poly_bits();
    put_words(w, final_words + 1);  // final_poly, pow_witness
    // public inputs: write_usize(len) (compat.bytes_pi_len_prefix) + elements
    const uint64_t n_pi = in->n_public_inputs;
    if (ctx->compat.bytes_pi_len_prefix) put_words(&n_pi, 1);
    put_words(in->public_inputs, in->n_public_inputs);
    return overflow ? (long)VPBS_ERR_INVALID : (long)pos;
}

}  // extern "C"
