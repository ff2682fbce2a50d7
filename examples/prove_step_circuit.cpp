// Pure C/C++ consumer of the C ABI for a circuit that arrives as data: the reference's step circuit without its recursive verifier
// (build_step_circuit, /root/reference/src/vtfhe/ivc_based_vpbs.rs:80-155), exported by tools/export_step_circuit.py in the shape the
// Rust side would export from CircuitData after builder.build() -- gates, gate per row, constants, copy constraints, gadget
// generators, the targets the PartialWitness sets and the public-input targets.  Steps, all through include/vpbs_prover.h:
//   vpbs_gates_layout, vpbs_sigma_values                                    (once per circuit)
//   vpbs_witness_plan_create                                               (once per circuit)
//   vpbs_witness_plan_run -> vpbs_check_witness                            (per proof, host)
//   vpbs_commit_values (constants + sigmas, once) -> vpbs_prove_step       (MI355X)
//   vpbs_verify_step                                                       (host)
// File format (little-endian u64 words): header {magic, log_n, n_wires, n_routed, n_gates, n_constants_cols, n_copies, n_generators,
// generator_words, n_preset, n_public_inputs}; gates [n_gates][kind, p0, p1, p2]; row_gate [n]; constants [cols][n]; copies [n_copies][2];
// generators {kind, p0, n_in, n_out, in.., out..}*; preset positions; public-input positions; sample preset values; expected public inputs;
// optional trailer {N, K, ELL, LOGB, n_lwe, used_rows}.
//   build: g++ -O2 -std=c++17 -I include examples/prove_step_circuit.cpp -L verifiable-fhe-paper_amd -lvpbs_hip \
//              -Wl,-rpath,$PWD/verifiable-fhe-paper_amd -o examples/prove_step_circuit
//   run  : python tools/export_step_circuit.py /tmp/step.bin 8 2 4 5 6 && examples/prove_step_circuit /tmp/step.bin
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vpbs_prover.h"

static std::vector<uint64_t> read_file(const char* path) {
    std::vector<uint64_t> out;
    FILE* f = std::fopen(path, "rb");
    if (!f) return out;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize((size_t)bytes / 8);
    if (std::fread(out.data(), 8, out.size(), f) != out.size()) out.clear();
    std::fclose(f);
    return out;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s circuit.bin\n", argv[0]);
        return 2;
    }
    const std::vector<uint64_t> file = read_file(argv[1]);
    if (file.size() < 11 || file[0] != 0x5354455043495243ull) {
        std::fprintf(stderr, "not a step-circuit file\n");
        return 2;
    }
    const unsigned log_n = (unsigned)file[1], n_wires = (unsigned)file[2], n_routed = (unsigned)file[3], n_gates = (unsigned)file[4],
                   n_const_cols = (unsigned)file[5];
    const size_t n_copies = file[6], n_generators = file[7], gen_words = file[8], n_preset = file[9], n_pi = file[10];
    const size_t n = (size_t)1 << log_n;
    const uint64_t* p = file.data() + 11;
    auto take = [&](size_t words) {
        const uint64_t* q = p;
        p += words;
        return q;
    };
    const uint64_t *f_gates = take(4 * n_gates), *f_rows = take(n), *f_consts = take((size_t)n_const_cols * n), *f_copies = take(2 * n_copies),
                   *f_gens = take(gen_words), *f_preset = take(n_preset), *f_pi = take(n_pi), *f_values = take(n_preset), *f_expect = take(n_pi);
    if ((size_t)(p - file.data()) + 6 == file.size()) p += 6;   // optional trailer {N, K, ELL, LOGB, n_lwe, used_rows}
    if ((size_t)(p - file.data()) + 8 == file.size()) {
        std::fprintf(stderr, "a cyclic / dummy circuit file carries no sample witness (it needs a proof): see tools/prove_ivc.py\n");
        return 2;
    }
    if ((size_t)(p - file.data()) != file.size()) {
        std::fprintf(stderr, "truncated file\n");
        return 2;
    }
    // ---- circuit description ----
    std::vector<vpbs_gate> gates(n_gates);
    for (unsigned i = 0; i < n_gates; ++i) {
        gates[i] = vpbs_gate{};
        gates[i].kind = (unsigned)f_gates[4 * i];
        gates[i].p0 = (unsigned)f_gates[4 * i + 1];
        gates[i].p1 = (unsigned)f_gates[4 * i + 2];
        gates[i].p2 = (unsigned)f_gates[4 * i + 3];
    }
    unsigned num_selectors = 0, num_gate_constraints = 0;
    if (vpbs_gates_layout(gates.data(), n_gates, 9, &num_selectors, &num_gate_constraints) != 0) return 1;
    std::vector<uint32_t> row_gate(f_rows, f_rows + n), copies(f_copies, f_copies + 2 * n_copies), preset_pos(f_preset, f_preset + n_preset);
    std::vector<std::vector<uint32_t>> gen_pos(n_generators);
    std::vector<vpbs_generator> gens(n_generators);
    const uint64_t* g = f_gens;
    for (size_t i = 0; i < n_generators; ++i) {
        gens[i].kind = (unsigned)g[0];
        gens[i].p0 = (unsigned)g[1];
        gens[i].n_in = (unsigned)g[2];
        gens[i].n_out = (unsigned)g[3];
        gen_pos[i].assign(g + 4, g + 4 + gens[i].n_in + gens[i].n_out);
        gens[i].in = gen_pos[i].data();
        gens[i].out = gen_pos[i].data() + gens[i].n_in;
        g += 4 + gens[i].n_in + gens[i].n_out;
    }
    vpbs_circuit circ{};
    circ.log_n = log_n; circ.n_wires = n_wires; circ.n_routed = n_routed;
    circ.gates = gates.data(); circ.n_gates = n_gates; circ.num_selectors = num_selectors;
    circ.row_gate = row_gate.data();
    circ.constants = f_consts; circ.n_constants_cols = n_const_cols;
    circ.copies = copies.data(); circ.n_copies = n_copies;
    circ.generators = gens.data(); circ.n_generators = n_generators;
    std::vector<uint64_t> sigma((size_t)n_routed * n), wires((size_t)n_wires * n);
    if (vpbs_sigma_values(&circ, sigma.data()) != 0) return 1;
    // ---- witness: compiled once, run per PartialWitness ----
    char err[256];
    vpbs_witness_plan* plan = nullptr;
    if (vpbs_witness_plan_create(&circ, preset_pos.data(), n_preset, &plan, err, sizeof err) != 0) {
        std::fprintf(stderr, "witness plan: %s\n", err);
        return 1;
    }
    if (vpbs_witness_plan_run(plan, f_values, 0, wires.data(), err, sizeof err) != 0) {
        std::fprintf(stderr, "witness generation failed: %s\n", err);
        return 1;
    }
    vpbs_witness_plan_free(plan);
    std::vector<uint64_t> pis(n_pi);
    for (size_t i = 0; i < n_pi; ++i) pis[i] = wires[f_pi[i]];
    size_t wrong = 0;
    for (size_t i = 0; i < n_pi; ++i) wrong += pis[i] != f_expect[i];
    uint64_t pi_hash[4];
    vpbs_hash_no_pad(pis.data(), n_pi, pi_hash);
    const int sat = vpbs_check_witness(&circ, wires.data(), pi_hash, err, sizeof err);
    if (wrong || sat != 1) {
        std::fprintf(stderr, "witness: %zu public inputs differ from the exported ones; constraints satisfied: %d %s\n", wrong, sat, err);
        return 1;
    }
    // ---- prove on the device ----
    vpbs_ctx* ctx = nullptr;
    if (vpbs_ctx_create(0, log_n < 10 ? 10 : log_n, 3, 4, &ctx) != 0) {
        std::fprintf(stderr, "no MI355X device / context creation failed\n");
        return 2;
    }
    std::vector<uint64_t> cs(f_consts, f_consts + (size_t)n_const_cols * n);
    cs.insert(cs.end(), sigma.begin(), sigma.end());
    vpbs_batch* cs_batch = nullptr;
    std::vector<uint64_t> cs_cap(64);
    if (vpbs_commit_values(ctx, cs.data(), n_const_cols + n_routed, log_n, &cs_batch, cs_cap.data()) != 0) {
        std::fprintf(stderr, "commit failed: %s\n", vpbs_last_error(ctx));
        return 1;
    }
    vpbs_step_inputs in{};
    in.log_n = log_n; in.n_wires = n_wires; in.n_zs_partial_products = 20; in.n_quotient = 16; in.num_challenges = 2;
    in.wires_values = wires.data();
    in.constants_sigmas = cs_batch;
    in.circuit_digest[0] = 1; in.circuit_digest[1] = 2; in.circuit_digest[2] = 3; in.circuit_digest[3] = 4;
    in.public_inputs = pis.data(); in.n_public_inputs = n_pi;
    in.forced_pow = VPBS_POW_ANY;
    in.sigmas_values = sigma.data();
    in.n_routed = n_routed; in.quotient_degree_factor = 8; in.n_constants = n_const_cols;
    in.gates = gates.data(); in.n_gates = n_gates; in.num_selectors = num_selectors;
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
    v.n_constants_sigmas = n_const_cols + n_routed; v.n_wires = n_wires; v.n_zs_partial_products = 20; v.n_quotient = 16;
    v.num_challenges = 2;
    v.constants_sigmas_cap = cs_cap.data();
    for (int i = 0; i < 4; ++i) v.circuit_digest[i] = in.circuit_digest[i];
    v.public_inputs = pis.data(); v.n_public_inputs = n_pi;
    v.fri_only = 0;   // full verification: vanishing identity at zeta with the gate constraints
    v.n_constants = n_const_cols; v.n_routed = n_routed; v.quotient_degree_factor = 8;
    v.gates = gates.data(); v.n_gates = n_gates; v.num_selectors = num_selectors;
    const int ok = vpbs_verify_step(&v, caps.data(), openings.data(), fri.data());
    std::vector<uint64_t> other(pis);
    other[n_pi / 2] ^= 1;
    v.public_inputs = other.data();
    const int ok_other = vpbs_verify_step(&v, caps.data(), openings.data(), fri.data());
    std::printf("step circuit: degree 2^%u, %zu copy constraints, %zu gadget generators, %zu public inputs\n", log_n, n_copies, n_generators, n_pi);
    std::printf("proof verified: %d; with a wrong public input: %d\n", ok, ok_other);
    vpbs_batch_free(cs_batch);
    vpbs_ctx_destroy(ctx);
    return ok == 1 && ok_other == 0 ? 0 : 1;
}
