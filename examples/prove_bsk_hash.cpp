// Pure C/C++ consumer of the C ABI: a circuit from description to verified proof, no Python, no torch.
//
// The circuit is the bootstrapping-key hash of the reference's step circuit
// (/root/reference/src/vtfhe/ivc_based_vpbs.rs:126-133): current_bsk_hash_out = hash_n_to_hash_no_pad(current_bsk_hash_in ||
// ggsw.flatten()), registered as public inputs -- PoseidonGate rows chained through copy constraints (overwrite-mode sponge), a
// PoseidonGate hashing the public inputs and the PublicInputGate, NoopGate padding.  Steps, all through include/vpbs_prover.h:
//   vpbs_gates_layout -> vpbs_selector_columns / vpbs_sigma_values (circuit build time)
//   vpbs_generate_witness (PartialWitness -> wires)            [host]
//   vpbs_commit_values (constants + sigmas, once) -> vpbs_prove_step with gates (wires commit, Z / partial products, gate
//   constraints + permutation quotient, openings, FRI)          [MI355X]
//   vpbs_verify_step (gate constraints re-evaluated at zeta)    [host]
// and the public inputs are compared with the native chain hash of verify_hash_output (ivc_based_vpbs.rs:64-78, vpbs_hash_chain).
//   build: g++ -O2 -std=c++17 -I include examples/prove_bsk_hash.cpp -L verifiable-fhe-paper_amd -lvpbs_hip \
//              -Wl,-rpath,$PWD/verifiable-fhe-paper_amd -o examples/prove_bsk_hash
//   run  : examples/prove_bsk_hash [K ELL N]   (default 2 4 1024: the paper's parameters, 2049 sponge rows, degree 2^12)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vpbs_prover.h"

static const uint64_t P = 0xFFFFFFFF00000001ull;

static std::vector<uint64_t> field_elements(uint64_t seed, size_t count) {  // splitmix64 stream, values >= p dropped
    std::vector<uint64_t> out;
    uint64_t state = seed;
    while (out.size() < count) {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if (z < P) out.push_back(z);
    }
    return out;
}

int main(int argc, char** argv) {
    const unsigned K = argc > 3 ? (unsigned)std::atoi(argv[1]) : 2, ELL = argc > 3 ? (unsigned)std::atoi(argv[2]) : 4,
                   N = argc > 3 ? (unsigned)std::atoi(argv[3]) : 1024;
    const unsigned n_wires = 135, n_routed = 80;
    // the bootstrapping-key element (Ggsw::flatten order) and the incoming chain hash
    const std::vector<uint64_t> item = field_elements(0xB5C, (size_t)K * ELL * K * N);
    std::vector<uint64_t> data(4, 0);  // current_bsk_hash_in = 0 (first CMUX step)
    data.insert(data.end(), item.begin(), item.end());
    const size_t n_chunks = (data.size() + 7) / 8;
    unsigned log_n = 3;
    while (((size_t)1 << log_n) < n_chunks + 2) ++log_n;
    const size_t n = (size_t)1 << log_n;

    // ---- gate set and layout (CircuitBuilder::build) ----
    vpbs_gate gates[3] = {};
    gates[0].kind = VPBS_GATE_NOOP;
    gates[1].kind = VPBS_GATE_PUBLIC_INPUT;
    gates[2].kind = VPBS_GATE_POSEIDON;
    unsigned num_selectors = 0, num_gate_constraints = 0;
    for (auto& g : gates)
        if (vpbs_gate_default_params(&g) != 0) return 1;
    if (vpbs_gates_layout(gates, 3, 9, &num_selectors, &num_gate_constraints) != 0) return 1;
    unsigned i_noop = 0, i_pi = 0, i_pos = 0;
    for (unsigned i = 0; i < 3; ++i) {
        if (gates[i].kind == VPBS_GATE_NOOP) i_noop = i;
        if (gates[i].kind == VPBS_GATE_PUBLIC_INPUT) i_pi = i;
        if (gates[i].kind == VPBS_GATE_POSEIDON) i_pos = i;
    }
    // ---- rows, copy constraints, partial witness ----
    std::vector<uint32_t> row_gate(n, i_noop);
    row_gate[0] = i_pi;
    for (size_t r = 1; r <= n_chunks + 1; ++r) row_gate[r] = i_pos;
    std::vector<uint32_t> copies, preset_pos;
    std::vector<uint64_t> preset_val;
    auto pos = [&](unsigned col, size_t row) { return (uint32_t)(col * n + row); };
    auto preset = [&](unsigned col, size_t row, uint64_t v) { preset_pos.push_back(pos(col, row)); preset_val.push_back(v); };
    auto copy = [&](uint32_t a, uint32_t b) { copies.push_back(a); copies.push_back(b); };
    for (size_t k = 0; k < n_chunks; ++k) {
        const size_t r = 1 + k, len = data.size() - 8 * k < 8 ? data.size() - 8 * k : 8;
        for (unsigned i = 0; i < 12; ++i) {
            if (i < len) preset(i, r, data[8 * k + i]);              // overwrite mode: the new block
            else if (k == 0) preset(i, r, 0);                         // initial sponge state
            else copy(pos(12 + i, r - 1), pos(i, r));                // the rest of the state carries over
        }
        preset(24, r, 0);  // swap
    }
    const size_t r_pi = 1 + n_chunks;
    for (unsigned i = 0; i < 12; ++i) {
        if (i < 4) {
            copy(pos(12 + i, r_pi - 1), pos(i, r_pi));  // public inputs = the hash output ...
            copy(pos(12 + i, r_pi), pos(i, 0));         // ... and hash_no_pad(public inputs) feeds the PublicInputGate
        } else preset(i, r_pi, 0);
    }
    preset(24, r_pi, 0);

    const unsigned n_constants = num_selectors + 1;  // one (unused) gate-constant column
    std::vector<uint64_t> constants((size_t)n_constants * n, 0);
    vpbs_circuit circ{};
    circ.log_n = log_n; circ.n_wires = n_wires; circ.n_routed = n_routed;
    circ.gates = gates; circ.n_gates = 3; circ.num_selectors = num_selectors;
    circ.row_gate = row_gate.data();
    circ.constants = constants.data(); circ.n_constants_cols = n_constants;
    circ.copies = copies.data(); circ.n_copies = copies.size() / 2;
    if (vpbs_selector_columns(&circ, constants.data()) != 0) return 1;
    std::vector<uint64_t> sigma((size_t)n_routed * n), wires((size_t)n_wires * n);
    if (vpbs_sigma_values(&circ, sigma.data()) != 0) return 1;
    char err[256];
    if (vpbs_generate_witness(&circ, preset_pos.data(), preset_val.data(), preset_pos.size(), wires.data(), err, sizeof err) != 0) {
        std::fprintf(stderr, "witness generation failed: %s\n", err);
        return 1;
    }
    uint64_t pis[4], native[4];
    for (unsigned i = 0; i < 4; ++i) pis[i] = wires[pos(12 + i, r_pi - 1)];
    if (vpbs_hash_chain(item.data(), 1, item.size(), pis, native) != 1) {
        std::fprintf(stderr, "in-circuit hash differs from the native chain hash\n");
        return 1;
    }

    // ---- prove on the device ----
    vpbs_ctx* ctx = nullptr;
    if (vpbs_ctx_create(0, 16, 3, 4, &ctx) != 0) {
        std::fprintf(stderr, "no MI355X device / context creation failed\n");
        return 2;
    }
    std::vector<uint64_t> cs(constants);
    cs.insert(cs.end(), sigma.begin(), sigma.end());
    vpbs_batch* cs_batch = nullptr;
    std::vector<uint64_t> cs_cap(64);
    if (vpbs_commit_values(ctx, cs.data(), n_constants + n_routed, log_n, &cs_batch, cs_cap.data()) != 0) {
        std::fprintf(stderr, "commit failed: %s\n", vpbs_last_error(ctx));
        return 1;
    }
    vpbs_step_inputs in{};
    in.log_n = log_n; in.n_wires = n_wires; in.n_zs_partial_products = 20; in.n_quotient = 16; in.num_challenges = 2;
    in.wires_values = wires.data();
    in.zs_pp_values = nullptr;      // Z / partial products on the device
    in.quotient_coeffs = nullptr;   // gate constraints + permutation quotient on the device
    in.constants_sigmas = cs_batch;
    in.circuit_digest[0] = 11; in.circuit_digest[1] = 22; in.circuit_digest[2] = 33; in.circuit_digest[3] = 44;
    in.public_inputs = pis; in.n_public_inputs = 4;
    in.forced_pow = VPBS_POW_ANY;
    in.sigmas_values = sigma.data();
    in.n_routed = n_routed; in.quotient_degree_factor = 8; in.n_constants = n_constants;
    in.gates = gates; in.n_gates = 3; in.num_selectors = num_selectors;
    vpbs_step_sizes sz{};
    if (vpbs_step_sizes_get(ctx, &in, &sz) != 0) return 1;
    std::vector<uint64_t> caps(3 * sz.cap_words), openings(sz.openings_words), fri(sz.fri_words);
    if (vpbs_prove_step(ctx, &in, caps.data(), openings.data(), fri.data(), nullptr, nullptr) != 0) {
        std::fprintf(stderr, "prove failed: %s\n", vpbs_last_error(ctx));
        return 1;
    }
    // ---- verify on the host ----
    vpbs_verify_inputs v{};
    v.log_n = log_n; v.rate_bits = 3; v.cap_height = 4;
    v.n_constants_sigmas = n_constants + n_routed; v.n_wires = n_wires; v.n_zs_partial_products = 20; v.n_quotient = 16;
    v.num_challenges = 2;
    v.constants_sigmas_cap = cs_cap.data();
    for (int i = 0; i < 4; ++i) v.circuit_digest[i] = in.circuit_digest[i];
    v.public_inputs = pis; v.n_public_inputs = 4;
    v.fri_only = 0;   // full verification: vanishing identity at zeta with the gate constraints
    v.n_constants = n_constants; v.n_routed = n_routed; v.quotient_degree_factor = 8;
    v.gates = gates; v.n_gates = 3; v.num_selectors = num_selectors;
    const int ok = vpbs_verify_step(&v, caps.data(), openings.data(), fri.data());
    uint64_t wrong[4] = {pis[0] ^ 1, pis[1], pis[2], pis[3]};
    v.public_inputs = wrong;
    const int ok_wrong = vpbs_verify_step(&v, caps.data(), openings.data(), fri.data());
    std::printf("bsk hash circuit: K=%u ELL=%u N=%u, %zu sponge rows, degree 2^%u, hash %016llx %016llx %016llx %016llx\n", K, ELL, N, n_chunks,
                log_n, (unsigned long long)pis[0], (unsigned long long)pis[1], (unsigned long long)pis[2], (unsigned long long)pis[3]);
    std::printf("proof verified: %d; with a wrong public input: %d\n", ok, ok_wrong);
    vpbs_batch_free(cs_batch);
    vpbs_ctx_destroy(ctx);
    return ok == 1 && ok_wrong == 0 ? 0 : 1;
}
