// Host-side verifier of step proofs (no device work): mirrors plonky2 0.2.0 plonk/verifier.rs `verify_with_challenges`,
// plonk/get_challenges.rs, fri/verifier.rs `verify_fri_proof` / `fri_verifier_query_round` / `fri_combine_initial` /
// `compute_evaluation`, hash/merkle_proofs.rs `verify_merkle_proof_to_cap` and plonk/vanishing_poly.rs
// `eval_vanishing_poly` (permutation part).  The reference calls it as `cd.verify(proof)` at
// /root/reference/src/vtfhe/ivc_based_vpbs.rs:443-447.  Product code: written against gl.h /
// poseidon.h, independent of the test oracle.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host/plonky2_mirror.h"
#include "kernels.h"
#include "poseidon.h"
#include "host/poseidon_x8.h"

using gl::Ext;
using gl::u64;

namespace {
void two_to_one(const u64* l, const u64* r, u64* out) {
    u64 s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    poseidon::permute_host(s);
    std::memcpy(out, s, 4 * sizeof(u64));
}
void hash_or_noop(const u64* leaf, size_t len, u64* out) {
    if (len <= 4) {
        std::memset(out, 0, 4 * sizeof(u64));
        std::memcpy(out, leaf, len * sizeof(u64));
    } else {
        poseidon::hash_no_pad_host(leaf, len, out);
    }
}
// verify_merkle_proof_to_cap
bool merkle_verify(const u64* leaf, size_t leaf_len, size_t idx, const u64* cap, const u64* siblings, size_t n_sib) {
    u64 cur[4], nxt[4];
    hash_or_noop(leaf, leaf_len, cur);
    for (size_t k = 0; k < n_sib; ++k) {
        if (idx & 1) two_to_one(siblings + 4 * k, cur, nxt);
        else two_to_one(cur, siblings + 4 * k, nxt);
        std::memcpy(cur, nxt, sizeof cur);
        idx >>= 1;
    }
    return std::memcmp(cur, cap + 4 * idx, sizeof cur) == 0;
}
// The Merkle checks of a proof are independent of each other and of the arithmetic checks: they are collected while the query rounds are
// replayed and verified together at the end -- eight paths of one shape (leaf length, path length) side by side, one per AVX-512 lane
// (host/poseidon_x8.h), where the CPU has them; one after the other otherwise.  4 oracles + 3 FRI rounds x 28 queries = 196 paths,
// ~3 200 permutations: two thirds of the verifier's time when done one by one.
struct MerkleJob {
    const u64* leaf;
    size_t leaf_len, idx;
    const u64* cap;
    const u64* siblings;
    size_t n_sib;
};
#if defined(VPBS_HAVE_POSEIDON_X8)
__attribute__((target("avx512f,avx512dq")))
bool merkle_verify_x8(const MerkleJob* jobs, unsigned cnt) {   // cnt <= 8 jobs of one shape
    using poseidon_x8::V;
    const size_t leaf_len = jobs[0].leaf_len, n_sib = jobs[0].n_sib;
    alignas(64) u64 lane[12][8];
    u64 cur[8][4];
    auto job = [&](unsigned l) -> const MerkleJob& { return jobs[l < cnt ? l : 0]; };   // spare lanes repeat the first job
    if (leaf_len <= 4) {   // hash_or_noop: padded, not hashed
        for (unsigned l = 0; l < 8; ++l)
            for (int i = 0; i < 4; ++i) cur[l][i] = (size_t)i < leaf_len ? job(l).leaf[i] : 0;
    } else {
        V st[12];
        for (int i = 0; i < 12; ++i) st[i] = _mm512_setzero_si512();
        for (size_t off = 0; off < leaf_len; off += 8) {   // overwrite-mode sponge, rate 8
            const size_t len = leaf_len - off < 8 ? leaf_len - off : 8;
            for (size_t i = 0; i < len; ++i) {
                for (unsigned l = 0; l < 8; ++l) lane[i][l] = job(l).leaf[off + i];
                st[i] = _mm512_load_si512(lane[i]);
            }
            poseidon_x8::permute(st, nullptr);
        }
        for (int i = 0; i < 4; ++i) {
            _mm512_store_si512(lane[i], st[i]);
            for (unsigned l = 0; l < 8; ++l) cur[l][i] = lane[i][l];
        }
    }
    size_t idx[8];
    for (unsigned l = 0; l < 8; ++l) idx[l] = job(l).idx;
    for (size_t k = 0; k < n_sib; ++k) {
        for (unsigned l = 0; l < 8; ++l) {
            const u64* sib = job(l).siblings + 4 * k;
            const bool right = idx[l] & 1;   // this node is the right child: two_to_one(sibling, node)
            for (int i = 0; i < 4; ++i) {
                lane[i][l] = right ? sib[i] : cur[l][i];
                lane[i + 4][l] = right ? cur[l][i] : sib[i];
            }
            idx[l] >>= 1;
        }
        V st[12];
        for (int i = 0; i < 8; ++i) st[i] = _mm512_load_si512(lane[i]);
        for (int i = 8; i < 12; ++i) st[i] = _mm512_setzero_si512();
        poseidon_x8::permute(st, nullptr);
        for (int i = 0; i < 4; ++i) {
            _mm512_store_si512(lane[i], st[i]);
            for (unsigned l = 0; l < 8; ++l) cur[l][i] = lane[i][l];
        }
    }
    for (unsigned l = 0; l < cnt; ++l)
        if (std::memcmp(cur[l], jobs[l].cap + 4 * idx[l], sizeof cur[l]) != 0) return false;
    return true;
}
#endif
bool merkle_verify_all(std::vector<MerkleJob>& jobs) {
#if defined(VPBS_HAVE_POSEIDON_X8)
    if (poseidon_x8::enabled()) {   // vpbs_host_set_poseidon_x8(0): one path after the other (A/B measurements, tests of that form)
        std::stable_sort(jobs.begin(), jobs.end(), [](const MerkleJob& a, const MerkleJob& b) {
            return a.leaf_len != b.leaf_len ? a.leaf_len < b.leaf_len : a.n_sib < b.n_sib;
        });
        for (size_t at = 0; at < jobs.size();) {
            size_t end = at + 1;
            while (end < jobs.size() && end - at < 8 && jobs[end].leaf_len == jobs[at].leaf_len && jobs[end].n_sib == jobs[at].n_sib) ++end;
            if (!merkle_verify_x8(jobs.data() + at, (unsigned)(end - at))) return false;
            at = end;
        }
        return true;
    }
#endif
    for (const MerkleJob& j : jobs)
        if (!merkle_verify(j.leaf, j.leaf_len, j.idx, j.cap, j.siblings, j.n_sib)) return false;
    return true;
}
Ext ext_at(const u64* p, size_t i) { return Ext{p[2 * i], p[2 * i + 1]}; }
size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
// interpolate {(xs[i], ys[i])} and evaluate at t (arity <= 16)
Ext interpolate(const u64* xs, const Ext* ys, size_t k, Ext t) {
    Ext res = gl::ext(0);
    for (size_t i = 0; i < k; ++i) {
        Ext num = ys[i];
        u64 den = 1;
        for (size_t j = 0; j < k; ++j)
            if (j != i) {
                num = gl::mul(num, gl::sub(t, gl::ext(xs[j])));
                den = gl::mul(den, gl::sub(xs[i], xs[j]));
            }
        res = gl::add(res, gl::mul(num, gl::inv(den)));
    }
    return res;
}

bool check_permutation_identity(const vpbs_verify_inputs* in, const u64* openings, const u64* betas, const u64* gammas,
                                const u64* alphas, Ext zeta, const u64* gate_terms_zeta) {
    const unsigned nc = in->num_challenges, n_routed = in->n_routed, deg = in->quotient_degree_factor;
    const unsigned n_chunks = (n_routed + deg - 1) / deg, num_prods = n_chunks - 1;
    const size_t n = (size_t)1 << in->log_n;
    const u64* cs_z = openings;
    const u64* wires_z = cs_z + 2 * (size_t)in->n_constants_sigmas;
    const u64* zs_pp_z = wires_z + 2 * (size_t)in->n_wires;
    const u64* quot_z = zs_pp_z + 2 * (size_t)in->n_zs_partial_products;
    const u64* zs_next_z = quot_z + 2 * (size_t)in->n_quotient;
    const u64* sig_z = cs_z + 2 * (size_t)in->n_constants;
    const u64* pps_z = zs_pp_z + 2 * (size_t)nc;
    Ext zeta_n = zeta;
    for (unsigned i = 0; i < in->log_n; ++i) zeta_n = gl::mul(zeta_n, zeta_n);
    const Ext one = gl::ext(1);
    const Ext z_h = gl::sub(zeta_n, one);
    const Ext l0 = gl::mul(z_h, gl::inv(gl::mul(gl::sub(zeta, one), (u64)n)));
    std::vector<Ext> terms(nc + nc * n_chunks);
    std::vector<u64> k_is(n_routed);
    u64 k = 1;
    for (unsigned j = 0; j < n_routed; ++j) { k_is[j] = k; k = gl::mul(k, gl::GENERATOR); }
    for (unsigned c = 0; c < nc; ++c) {
        terms[c] = gl::mul(l0, gl::sub(ext_at(zs_pp_z, c), one));
        for (unsigned kk = 0; kk < n_chunks; ++kk) {
            Ext num = one, den = one;
            for (unsigned j = kk * deg; j < (kk + 1) * deg && j < n_routed; ++j) {
                const Ext w = ext_at(wires_z, j), g = gl::ext(gammas[c]);
                num = gl::mul(num, gl::add(gl::add(w, gl::mul(zeta, gl::mul(betas[c], k_is[j]))), g));
                den = gl::mul(den, gl::add(gl::add(w, gl::mul(ext_at(sig_z, j), betas[c])), g));
            }
            const Ext prev = kk == 0 ? ext_at(zs_pp_z, c) : ext_at(pps_z, c * num_prods + kk - 1);
            const Ext next = kk == num_prods ? ext_at(zs_next_z, c) : ext_at(pps_z, c * num_prods + kk);
            terms[nc + c * n_chunks + kk] = gl::sub(gl::mul(prev, num), gl::mul(next, den));  // check_partial_products
        }
    }
    const unsigned chunks_per = 1u << in->rate_bits;
    for (unsigned a = 0; a < nc; ++a) {
        Ext acc = gate_terms_zeta ? ext_at(gate_terms_zeta, a) : gl::ext(0);
        for (size_t i = terms.size(); i-- > 0;) acc = gl::add(gl::mul(acc, alphas[a]), terms[i]);  // reduce_with_powers
        Ext q = gl::ext(0);
        for (unsigned m = chunks_per; m-- > 0;) q = gl::add(gl::mul(q, zeta_n), ext_at(quot_z, a * chunks_per + m));
        if (!gl::eq(acc, gl::mul(z_h, q))) return false;
    }
    return true;
}
}  // namespace

extern "C" int vpbs_verify_step(const vpbs_verify_inputs* in, const uint64_t* caps, const uint64_t* openings, const uint64_t* fri) {
    using namespace plonky2;
    if (!in || !caps || !openings || !fri || !in->constants_sigmas_cap || (in->n_public_inputs && !in->public_inputs)) return VPBS_ERR_INVALID;
    if (in->rate_bits > 3 || in->cap_height > 8 || in->log_n == 0 || in->log_n + in->rate_bits > 24 || in->num_challenges == 0 ||
        in->num_challenges > in->n_zs_partial_products)
        return VPBS_ERR_INVALID;
    if (!in->fri_only) {
        const unsigned deg = in->quotient_degree_factor;
        if (deg == 0 || in->n_routed == 0 || in->n_routed > in->n_wires || in->n_constants + in->n_routed > in->n_constants_sigmas ||
            in->n_zs_partial_products != in->num_challenges * ((in->n_routed + deg - 1) / deg) ||
            in->n_quotient != (in->num_challenges << in->rate_bits))
            return VPBS_ERR_INVALID;
    }
    const FriParams fp = FriParams::standard(in->log_n, in->rate_bits, in->cap_height);
    if (!fp.caps_fit()) return VPBS_ERR_INVALID;
    vpbs_compat compat;   // the switch table of include/vpbs_prover.h (NULL = plonky2 0.2.0 as restated)
    vpbs_compat_default(&compat);
    if (in->compat) compat = *in->compat;
    const unsigned nc = in->num_challenges, log_n = in->log_n;
    const unsigned log_lde = log_n + in->rate_bits;
    const size_t lde = (size_t)1 << log_lde, cap_words = (size_t)4 << in->cap_height;
    const size_t ncols[4] = {in->n_constants_sigmas, in->n_wires, in->n_zs_partial_products, in->n_quotient};
    size_t total_cols = 0;
    for (size_t c : ncols) total_cols += c;

    // ---- transcript (plonk/get_challenges.rs) ----
    HashOut pi_hash;
    vpbs_hash_no_pad(in->public_inputs, in->n_public_inputs, pi_hash.data());
    Challenger ch;
    ch.observe_elements(in->circuit_digest, 4);
    ch.observe_hash(pi_hash);
    ch.observe_cap(caps, cap_words / 4);
    const std::vector<u64> betas = ch.get_n_challenges(nc), gammas = ch.get_n_challenges(nc);
    ch.observe_cap(caps + cap_words, cap_words / 4);
    const std::vector<u64> alphas = ch.get_n_challenges(nc);
    ch.observe_cap(caps + 2 * cap_words, cap_words / 4);
    const Ext zeta = ch.get_extension_challenge();
    ch.observe_elements(openings, 2 * (total_cols + nc));
    if (!in->fri_only) {
        // eval_vanishing_poly: the gate constraints at zeta come from the openings of the constants and the wires
        const u64* gate_terms = in->gate_terms_zeta;
        std::vector<u64> gt(2 * (size_t)nc);
        if (in->gates && in->n_gates) {
            if (in->num_selectors > in->n_constants) return VPBS_ERR_INVALID;
            try {
                vpbs::validate_gates(in->gates, in->n_gates, in->num_selectors, in->n_constants, in->n_wires);
            } catch (const vpbs::DeviceError&) {
                return VPBS_ERR_INVALID;
            }
            vpbs::gate_terms_at(in->gates, in->n_gates, in->num_selectors, openings, in->n_constants, openings + 2 * (size_t)in->n_constants_sigmas,
                                in->n_wires, pi_hash.data(), alphas.data(), nc, gt.data());
            gate_terms = gt.data();
        }
        if (!check_permutation_identity(in, openings, betas.data(), gammas.data(), alphas.data(), zeta, gate_terms)) return 0;
    }

    // ---- FRI challenges ----
    const Ext fri_alpha = ch.get_extension_challenge();
    const size_t n_rounds = fp.reduction_arity_bits.size();
    const size_t final_len = (size_t)1 << fp.final_poly_bits();
    const size_t total = fri_proof_words(fp, {ncols[0], ncols[1], ncols[2], ncols[3]});
    const u64* final_words = fri + total - 1 - 2 * final_len;
    const u64 pow_witness = fri[total - 1];
    const u64* w = fri;
    std::vector<const u64*> fri_caps(n_rounds);
    std::vector<Ext> fri_betas(n_rounds);
    for (size_t r = 0; r < n_rounds; ++r) {
        fri_caps[r] = w;
        ch.observe_cap(w, cap_words / 4);
        w += cap_words;
        fri_betas[r] = ch.get_extension_challenge();
    }
    ch.observe_elements(final_words, 2 * final_len);
    if (pow_witness >= gl::P) return 0;
    ch.observe_element(pow_witness);
    const u64 pow_response = ch.get_challenge();
    if (fp.config.proof_of_work_bits && (pow_response >> (64 - fp.config.proof_of_work_bits)) != 0) return 0;

    // ---- PrecomputedReducedOpenings: batch 0 = every polynomial at zeta, batch 1 = Z polynomials at g * zeta ----
    const Ext zeta_next = gl::mul(zeta, gl::root_of_unity(log_n));
    Ext reduced0 = gl::ext(0), reduced1 = gl::ext(0);
    for (size_t j = total_cols; j-- > 0;) reduced0 = gl::add(gl::mul(reduced0, fri_alpha), ext_at(openings, j));
    for (size_t j = nc; j-- > 0;) reduced1 = gl::add(gl::mul(reduced1, fri_alpha), ext_at(openings, total_cols + j));
    const u64* oracle_caps[4] = {in->constants_sigmas_cap, caps, caps + cap_words, caps + 2 * cap_words};

    std::vector<MerkleJob> merkle_jobs;
    merkle_jobs.reserve((size_t)fp.config.num_query_rounds * (4 + n_rounds));
    for (unsigned q = 0; q < fp.config.num_query_rounds; ++q) {
        size_t x_index = (size_t)(ch.get_challenge() % lde);
        const u64* leaf[4];
        const size_t nsib0 = log_lde - in->cap_height;
        for (size_t o = 0; o < 4; ++o) {  // fri_verify_initial_proof
            leaf[o] = w;
            merkle_jobs.push_back({w, ncols[o], x_index, oracle_caps[o], w + ncols[o], nsib0});
            w += ncols[o] + 4 * nsib0;
        }
        u64 subgroup_x = gl::mul(gl::GENERATOR, gl::pow(gl::root_of_unity(log_lde), bitrev(x_index, log_lde)));
        // fri_combine_initial
        Ext sum = gl::ext(0);
        {
            Ext acc = gl::ext(0), apow = gl::ext(1);
            for (size_t o = 0; o < 4; ++o)
                for (size_t p = 0; p < ncols[o]; ++p) {
                    acc = gl::add(acc, gl::mul(apow, leaf[o][p]));
                    apow = gl::mul(apow, fri_alpha);
                }
            sum = gl::mul(gl::sub(acc, reduced0), gl::inv(gl::sub(gl::ext(subgroup_x), zeta)));
            acc = gl::ext(0);
            apow = gl::ext(1);
            for (size_t p = 0; p < nc; ++p) {
                acc = gl::add(acc, gl::mul(apow, leaf[2][p]));
                apow = gl::mul(apow, fri_alpha);
            }
            sum = gl::add(gl::mul(sum, apow), gl::mul(gl::sub(acc, reduced1), gl::inv(gl::sub(gl::ext(subgroup_x), zeta_next))));
            // compat.fri_mul_final_by_x: the prover multiplied the final polynomial by X, so the combined value carries a factor subgroup_x
            if (compat.fri_mul_final_by_x) sum = gl::mul(sum, subgroup_x);
        }
        Ext old_eval = sum;
        unsigned lg = log_lde;
        for (size_t r = 0; r < n_rounds; ++r) {
            const unsigned ab = fp.reduction_arity_bits[r];
            const size_t arity = (size_t)1 << ab;
            const u64* evals = w;
            const size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (evals[2 * within] != old_eval.c0 || evals[2 * within + 1] != old_eval.c1) return 0;
            // compute_evaluation
            const u64 g = gl::root_of_unity(ab);
            const u64 coset_start = gl::mul(subgroup_x, gl::pow(g, arity - bitrev(within, ab)));
            u64 xs[16];
            Ext ys[16];
            u64 y = 1;
            for (size_t i = 0; i < arity; ++i) {
                xs[i] = gl::mul(coset_start, y);
                ys[i] = ext_at(evals, bitrev(i, ab));
                y = gl::mul(y, g);
            }
            old_eval = interpolate(xs, ys, arity, fri_betas[r]);
            lg -= ab;
            const size_t nsib = lg - in->cap_height;
            merkle_jobs.push_back({evals, 2 * arity, coset_index, fri_caps[r], evals + 2 * arity, nsib});
            w += 2 * arity + 4 * nsib;
            for (unsigned k = 0; k < ab; ++k) subgroup_x = gl::mul(subgroup_x, subgroup_x);
            x_index = coset_index;
        }
        Ext acc = gl::ext(0);
        for (size_t i = final_len; i-- > 0;) acc = gl::add(gl::mul(acc, subgroup_x), ext_at(final_words, i));
        if (!gl::eq(acc, old_eval)) return 0;
    }
    return merkle_verify_all(merkle_jobs) ? 1 : 0;
}

// The inverse of vpbs_step_proof_to_bytes (prover.hip): ProofWithPublicInputs bytes -> the flat arrays vpbs_verify_step takes.  The
// shape comes from `in` (column counts, n_constants, degree); a byte string of another shape is rejected, not guessed at.
extern "C" long vpbs_step_proof_from_bytes(const vpbs_verify_inputs* in, const uint8_t* bytes, size_t len, uint64_t* caps, uint64_t* openings, uint64_t* fri,
                                uint64_t* public_inputs_out, size_t public_inputs_capacity) {
    if (!in || !bytes || !caps || !openings || !fri || in->n_constants > in->n_constants_sigmas || in->num_challenges > in->n_zs_partial_products)
        return VPBS_ERR_INVALID;
    using namespace plonky2;
    if (in->rate_bits > 3 || in->cap_height > 8 || in->log_n == 0 || in->log_n + in->rate_bits > 24) return VPBS_ERR_INVALID;
    const FriParams fp = FriParams::standard(in->log_n, in->rate_bits, in->cap_height);
    if (!fp.caps_fit()) return VPBS_ERR_INVALID;
    size_t pos = 0;
    bool bad = false;
    auto get_words = [&](uint64_t* w, size_t cnt) {
        if (bad || pos + 8 * cnt > len) {
            bad = true;
            return;
        }
        std::memcpy(w, bytes + pos, 8 * cnt);
        for (size_t i = 0; i < cnt; ++i)
            if (w[i] >= gl::P) bad = true;  // non-canonical field element
        pos += 8 * cnt;
    };
    auto expect_u8 = [&](unsigned v) {
        if (bad || pos + 1 > len || bytes[pos] != (uint8_t)v) bad = true;
        ++pos;
    };
    const size_t cap_words = (size_t)4 << in->cap_height;
    const size_t n_cs = in->n_constants_sigmas, nc = in->num_challenges;
    get_words(caps, 3 * cap_words);
    uint64_t* cs = openings;
    uint64_t* wires = cs + 2 * n_cs;
    uint64_t* zs_pp = wires + 2 * (size_t)in->n_wires;
    uint64_t* quot = zs_pp + 2 * (size_t)in->n_zs_partial_products;
    uint64_t* zs_next = quot + 2 * (size_t)in->n_quotient;
    get_words(cs, 2 * (size_t)in->n_constants);
    get_words(cs + 2 * (size_t)in->n_constants, 2 * (n_cs - in->n_constants));
    get_words(wires, 2 * (size_t)in->n_wires);
    get_words(zs_pp, 2 * nc);
    get_words(zs_next, 2 * nc);
    get_words(zs_pp + 2 * nc, 2 * (size_t)(in->n_zs_partial_products - nc));
    get_words(quot, 2 * (size_t)in->n_quotient);
    uint64_t* w = fri;
    const size_t n_rounds = fp.reduction_arity_bits.size();
    get_words(w, n_rounds * cap_words);
    w += n_rounds * cap_words;
    const unsigned log_lde = fp.lde_bits();
    const size_t oracle_cols[4] = {n_cs, in->n_wires, in->n_zs_partial_products, in->n_quotient};
    for (unsigned q = 0; q < fp.config.num_query_rounds && !bad; ++q) {
        for (size_t o = 0; o < 4; ++o) {
            const unsigned nsib = log_lde - fp.config.cap_height;
            get_words(w, oracle_cols[o]);
            w += oracle_cols[o];
            expect_u8(nsib);
            get_words(w, 4 * (size_t)nsib);
            w += 4 * (size_t)nsib;
        }
        unsigned lg = log_lde;
        for (unsigned ab : fp.reduction_arity_bits) {
            lg -= ab;
            const unsigned nsib = lg - fp.config.cap_height;
            get_words(w, (size_t)2 << ab);
            w += (size_t)2 << ab;
            expect_u8(nsib);
            get_words(w, 4 * (size_t)nsib);
            w += 4 * (size_t)nsib;
        }
    }
    const size_t final_words = (size_t)2 << fp.final_poly_bits();
    get_words(w, final_words);
    if (!bad && pos + 8 <= len) {  // pow_witness: a plain u64, not a field element
        std::memcpy(w + final_words, bytes + pos, 8);
        pos += 8;
    } else {
        bad = true;
    }
    uint64_t n_pi = 0;
    const bool pi_prefix = in->compat ? in->compat->bytes_pi_len_prefix != 0 : true;   // the switch table of include/vpbs_prover.h
    if (!pi_prefix) {   // older layout: the public inputs run to the end of the buffer
        if (!bad && (len - pos) % 8 == 0) n_pi = (len - pos) / 8;
        else bad = true;
    } else if (!bad && pos + 8 <= len) {
        std::memcpy(&n_pi, bytes + pos, 8);
        pos += 8;
    } else {
        bad = true;
    }
    if (bad || n_pi > public_inputs_capacity || (n_pi && !public_inputs_out)) return VPBS_ERR_INVALID;
    get_words(public_inputs_out, (size_t)n_pi);
    if (bad || pos != len) return VPBS_ERR_INVALID;
    return (long)n_pi;
}


// verify_pbs of the reference (/root/reference/src/vtfhe/ivc_based_vpbs.rs:388-489), check for check and in its order: the claimed test
// vector, the step counter, the output ciphertext, cd.verify(proof) (vpbs_verify_step, full check), check_cyclic_proof_verifier_data (the
// proof's last public inputs are the circuit's own digest and constants/sigmas cap), verify_hash_output over the bootstrapping / key
// switching keys and over the LWE masks.  The reference panics at the first failing check; here the verdict is returned and `why` names it.
extern "C" int vpbs_verify_pbs(const vpbs_verify_pbs_inputs* in, const uint8_t* proof_bytes, size_t len, char* why, size_t why_len) {
    auto say = [&](const char* m) {
        if (why && why_len) {
            std::strncpy(why, m, why_len - 1);
            why[why_len - 1] = 0;
        }
    };
    say("");
    // out_ct is part of the statement (the reference asserts it, :440-442): a verdict without it would not bind the proof to the ciphertext
    // the caller holds.  ggsw_len strides the caller's bsk / ksk arrays: it must be a whole number of [K][N] GLWE rows (K * ELL * K * N)
    if (!in || !in->circuit || !proof_bytes || !in->testv || !in->out_ct || !in->ct || !in->ksk || (in->n_lwe && !in->bsk) || in->N == 0 ||
        in->K == 0 || in->ggsw_len == 0 || in->ggsw_len % ((size_t)in->K * in->K * in->N) != 0) {
        say("malformed arguments (testv, out_ct, ct, ksk, bsk are all required; ggsw_len = K * ELL * K * N)");
        return VPBS_ERR_INVALID;
    }
    const vpbs_verify_inputs& c = *in->circuit;
    const size_t kn = (size_t)in->K * in->N, cap_words = (size_t)4 << c.cap_height;
    const size_t n_pi = 2 * kn + 1 + 8 + 4 + cap_words;   // acc_init | counter | acc | two hashes | verifier data (digest, cap)
    std::vector<uint64_t> caps(3 * cap_words),
        openings(2 * ((size_t)c.n_constants_sigmas + c.n_wires + c.n_zs_partial_products + c.n_quotient + c.num_challenges)), fri(len / 8 + 8), pis(n_pi);
    if (vpbs_step_proof_from_bytes(&c, proof_bytes, len, caps.data(), openings.data(), fri.data(), pis.data(), n_pi) != (long)n_pi) {
        say("the bytes are not a proof of this circuit (shape, canonical field elements, number of public inputs)");
        return 0;
    }
    // claimed test vector: K - 1 zero polynomials, then testv (:422-433)
    for (size_t i = 0; i < kn - in->N; ++i)
        if (pis[i] != 0) {
            say("claimed test vector: the mask polynomials are not zero");
            return 0;
        }
    if (std::memcmp(pis.data() + kn - in->N, in->testv, 8 * (size_t)in->N) != 0) {
        say("claimed test vector differs from testv");
        return 0;
    }
    if (pis[kn] != (uint64_t)in->n_lwe + 2) {   // :435-438
        say("the counter is not n + 2");
        return 0;
    }
    if (std::memcmp(pis.data() + kn + 1, in->out_ct, 8 * kn) != 0) {   // :440-442
        say("the output ciphertext is not the proof's accumulator");
        return 0;
    }
    vpbs_verify_inputs v = c;   // cd.verify (:444-448)
    v.public_inputs = pis.data();
    v.n_public_inputs = n_pi;
    v.fri_only = 0;
    const int ok = vpbs_verify_step(&v, caps.data(), openings.data(), fri.data());
    if (ok < 0) {
        say("malformed circuit description");
        return ok;
    }
    if (ok != 1) {
        say("the proof does not verify");
        return 0;
    }
    // check_cyclic_proof_verifier_data (:449-453): the verifier data the chain was run with is this circuit's
    const uint64_t* vk = pis.data() + n_pi - 4 - cap_words;
    if (std::memcmp(vk, c.circuit_digest, 32) != 0 || !c.constants_sigmas_cap || std::memcmp(vk + 4, c.constants_sigmas_cap, 8 * cap_words) != 0) {
        say("the proof carries another circuit's verifier data");
        return 0;
    }
    // verify_hash_output (:454-481): dummy GGSW, the n bootstrapping keys, the key-switching key / ct[n], the n masks, zero
    const size_t steps = (size_t)in->n_lwe + 2, g = in->ggsw_len;
    {
        std::vector<uint64_t> items(steps * g, 0);
        if (in->n_lwe) std::memcpy(items.data() + g, in->bsk, 8 * (size_t)in->n_lwe * g);
        std::memcpy(items.data() + (steps - 1) * g, in->ksk, 8 * g);
        if (vpbs_hash_chain(items.data(), steps, g, pis.data() + 2 * kn + 1, nullptr) != 1) {
            say("the key hash chain does not match");
            return 0;
        }
    }
    std::vector<uint64_t> masks(steps, 0);
    masks[0] = in->ct[in->n_lwe];
    for (unsigned i = 0; i < in->n_lwe; ++i) masks[i + 1] = in->ct[i];
    if (vpbs_hash_chain(masks.data(), steps, 1, pis.data() + 2 * kn + 5, nullptr) != 1) {
        say("the LWE hash chain does not match");
        return 0;
    }
    return 1;
}
