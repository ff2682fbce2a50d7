// One verifiable PBS as the reference produces it -- an IVC chain (/root/reference/src/vtfhe/ivc_based_vpbs.rs:159-386 `verified_pbs`,
// :388-489 `verify_pbs`) -- from a plain C++ host through the C ABI of include/vpbs_prover.h: no Python, no HIP runtime in this translation
// unit.  This is the shape of the Rust side's main.rs: the cyclic step circuit and its dummy circuit arrive as data (the
// exports of tools/export_step_circuit.py --cyclic, standing in for what CircuitBuilder::build leaves behind), then
//     vpbs_ivc_create     commitments, verifier data, compiled + split witness plans, wire matrices
//     vpbs_keygen / vpbs_testv / vpbs_lwe_encrypt       main.rs:40-52 with seeded generators, the paper's noise
//     vpbs_ivc_prove_pbs  base proof + n + 2 chained step proofs (witness phases and uploads pipelined beside the proofs) -> the last proof's bytes
//     vpbs_verify_pbs     the reference's verify_pbs on that one proof;  vpbs_glwe_decrypt: the bootstrapped ciphertext decrypts to the message
//   build: g++ -O2 -std=c++17 -pthread -I include examples/prove_ivc.cpp -L verifiable-fhe-paper_amd -lvpbs_hip
//              -Wl,-rpath,$PWD/verifiable-fhe-paper_amd -o examples/prove_ivc
//   run  : python tools/export_step_circuit.py --cyclic /tmp/cyc.bin /tmp/dum.bin 8 2 4 5 6 13 && examples/prove_ivc /tmp/cyc.bin /tmp/dum.bin
//          examples/prove_ivc CYCLIC.bin DUMMY.bin [steps]        (steps < n + 2: a prefix of the chain; verify_pbs then does not apply)
//          VPBS_IVC_DEVICE_WITNESS=64 examples/prove_ivc ...       the early witness phases on the device, 64 steps per batch
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vpbs_prover.h"

namespace {
using u64 = uint64_t;
constexpr u64 P = 0xFFFFFFFF00000001ull;

#define REQUIRE(cond, ...)                       \
    do {                                         \
        if (!(cond)) {                           \
            std::fprintf(stderr, __VA_ARGS__);   \
            std::fprintf(stderr, "\n");          \
            std::exit(1);                        \
        }                                        \
    } while (0)

std::vector<u64> read_file(const char* path) {
    std::vector<u64> out;
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

// an exported circuit (verifiable-fhe-paper_amd/circuit_file.py documents the format) on one context
struct Circuit {
    std::vector<u64> file;
    unsigned log_n = 0, n_wires = 0, n_routed = 0, n_gates = 0, n_const_cols = 0, num_selectors = 0;
    size_t n = 0, n_preset = 0, n_pi = 0;
    u64 meta[8] = {0};
    std::vector<vpbs_gate> gates;
    std::vector<uint32_t> row_gate, copies, preset_pos, pi_pos;
    std::vector<std::vector<uint32_t>> gen_pos;
    std::vector<vpbs_generator> gens;
    vpbs_circuit circ{};

    void load(const char* path) {
        file = read_file(path);
        REQUIRE(file.size() > 11 && file[0] == 0x5354455043495243ull, "%s: not a circuit file", path);
        log_n = (unsigned)file[1]; n_wires = (unsigned)file[2]; n_routed = (unsigned)file[3]; n_gates = (unsigned)file[4];
        n_const_cols = (unsigned)file[5];
        const size_t n_copies = file[6], n_generators = file[7], gen_words = file[8];
        n_preset = file[9]; n_pi = file[10];
        n = (size_t)1 << log_n;
        const u64* p = file.data() + 11;
        auto take = [&](size_t words) { const u64* q = p; p += words; return q; };
        const u64 *f_gates = take(4 * n_gates), *f_rows = take(n), *f_consts = take((size_t)n_const_cols * n), *f_copies = take(2 * n_copies),
                  *f_gens = take(gen_words), *f_preset = take(n_preset), *f_pi = take(n_pi);
        take(n_preset); take(n_pi);   // sample sections (zero in cyclic / dummy files)
        const size_t left = file.size() - (size_t)(p - file.data());
        REQUIRE(left == 8, "%s: not a cyclic / dummy circuit file (trailer of %zu words)", path, left);
        std::memcpy(meta, p, sizeof meta);   // N, K, ELL, LOGB, n_lwe, used_rows, kind, proof_words
        gates.assign(n_gates, vpbs_gate{});
        for (unsigned i = 0; i < n_gates; ++i) {
            gates[i].kind = (unsigned)f_gates[4 * i]; gates[i].p0 = (unsigned)f_gates[4 * i + 1];
            gates[i].p1 = (unsigned)f_gates[4 * i + 2]; gates[i].p2 = (unsigned)f_gates[4 * i + 3];
        }
        unsigned ngc = 0;
        REQUIRE(vpbs_gates_layout(gates.data(), n_gates, 9, &num_selectors, &ngc) == 0, "gate layout failed");
        row_gate.assign(f_rows, f_rows + n);
        copies.assign(f_copies, f_copies + 2 * n_copies);
        preset_pos.assign(f_preset, f_preset + n_preset);
        pi_pos.assign(f_pi, f_pi + n_pi);
        gen_pos.resize(n_generators);
        gens.resize(n_generators);
        const u64* g = f_gens;
        for (size_t i = 0; i < n_generators; ++i) {
            gens[i].kind = (unsigned)g[0]; gens[i].p0 = (unsigned)g[1]; gens[i].n_in = (unsigned)g[2]; gens[i].n_out = (unsigned)g[3];
            gen_pos[i].assign(g + 4, g + 4 + gens[i].n_in + gens[i].n_out);
            gens[i].in = gen_pos[i].data();
            gens[i].out = gen_pos[i].data() + gens[i].n_in;
            g += 4 + gens[i].n_in + gens[i].n_out;
        }
        circ.log_n = log_n; circ.n_wires = n_wires; circ.n_routed = n_routed;
        circ.gates = gates.data(); circ.n_gates = n_gates; circ.num_selectors = num_selectors;
        circ.row_gate = row_gate.data();
        circ.constants = f_consts; circ.n_constants_cols = n_const_cols;
        circ.copies = copies.data(); circ.n_copies = n_copies;
        circ.generators = gens.data(); circ.n_generators = n_generators;
    }
    vpbs_ivc_circuit describe(size_t proof_words) const { return {&circ, preset_pos.data(), n_preset, pi_pos.data(), n_pi, proof_words}; }
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char** argv) {
    REQUIRE(argc >= 3, "usage: %s CYCLIC.bin DUMMY.bin [steps]", argv[0]);
    const std::vector<u64> head = read_file(argv[1]);
    REQUIRE(head.size() > 11 && head[1] >= 3 && head[1] <= 20, "%s: not a circuit file", argv[1]);
    vpbs_ctx* ctx = nullptr;
    REQUIRE(vpbs_ctx_create(0, std::max(16u, (unsigned)head[1]), 3, 4, &ctx) == 0, "no MI355X device / context creation failed");
    Circuit cyc, dum;
    cyc.load(argv[1]);
    dum.load(argv[2]);
    const unsigned N = (unsigned)cyc.meta[0], K = (unsigned)cyc.meta[1], ELL = (unsigned)cyc.meta[2], LOGB = (unsigned)cyc.meta[3],
                   n_lwe = (unsigned)cyc.meta[4];
    const size_t proof_words = cyc.meta[7], kn = (size_t)K * N, ggsw_len = (size_t)K * ELL * K * N, n_pi = cyc.n_pi;
    REQUIRE(cyc.meta[6] == 1 && dum.meta[6] == 2, "not a cyclic / dummy pair");
    const unsigned total = n_lwe + 2;
    const unsigned steps = argc > 3 ? (unsigned)std::min<long>(total, std::atol(argv[3])) : total;
    unsigned log_N = 0;
    while ((1u << log_N) < N) ++log_N;
    char err[256];
    vpbs_ivc* ivc = nullptr;
    const vpbs_ivc_circuit c_desc = cyc.describe(proof_words), d_desc = dum.describe(0);
    // one chain in this process: with 16 CPUs or more, 14 threads for the late witness phase (its last stage is 28 independent FRI queries)
    if (vpbs_host_cpu_budget() >= 16) vpbs_host_set_late_threads(14);
    REQUIRE(vpbs_ivc_create(ctx, &c_desc, &d_desc, N, K, ggsw_len, /* comm: one GPU */ nullptr, &ivc, err, sizeof err) == 0, "vpbs_ivc_create: %s", err);
    std::vector<u64> vk(68);
    vpbs_ivc_verifier_data(ivc, vk.data(), nullptr);
    // VPBS_IVC_DEVICE_WITNESS=b: the early witness phases of b steps at a time on the device (the host keeps the late phase); same proof
    if (const char* dw = std::getenv("VPBS_IVC_DEVICE_WITNESS"))
        REQUIRE(vpbs_ivc_set_device_witness(ivc, ELL, LOGB, (unsigned)std::atoi(dw), std::getenv("VPBS_IVC_DEVICE_LATE") != nullptr) == 0,
                "vpbs_ivc_set_device_witness failed");

    // ---- main.rs:40-52 with seeded generators ----
    vpbs_keygen_params kp{log_N, K, ELL, LOGB, n_lwe, 0x5EED0728ull, 4.99027217501041e-8, 1.17021618159313e-5};
    std::vector<u64> s_lwe(n_lwe), s_glwe((size_t)(K - 1) * N), s_to((size_t)K * N), bsk((size_t)n_lwe * ggsw_len), ksk(ggsw_len), testv(N), ct(n_lwe + 1);
    REQUIRE(vpbs_keygen(ctx, &kp, s_lwe.data(), s_glwe.data(), s_to.data(), bsk.data(), ksk.data(), 0) == 0, "keygen: %s", vpbs_last_error(ctx));
    u64 delta = 0;
    REQUIRE(vpbs_testv(log_N, 2, testv.data(), &delta) == 0, "testv");
    const u64 message = 1;
    REQUIRE(vpbs_lwe_encrypt(&kp, s_lwe.data(), delta * message % P, 0, ct.data()) == 0, "lwe_encrypt");

    // ---- verified_pbs (:159-386) ----
    std::vector<uint8_t> bytes(8 * (proof_words + n_pi) + 8192);
    vpbs_ivc_timing t{};
    const bool timing = std::getenv("VPBS_TIMING") != nullptr;   // HIP events around every kernel group of the step proofs
    if (timing) vpbs_timing_enable(ctx, 1);
    const long n_bytes = vpbs_ivc_prove_pbs(ivc, testv.data(), ct.data(), bsk.data(), ksk.data(), n_lwe, steps, bytes.data(), bytes.size(), &t, err, sizeof err);
    REQUIRE(n_bytes > 0, "vpbs_ivc_prove_pbs: %s", err);
    if (timing) {
        std::vector<char> report(1 << 14);
        vpbs_timing_report(ctx, report.data(), report.size());
        std::printf("device time of the base proof + %u step proofs by kernel group: %s\n", steps, report.data());
        vpbs_timing_enable(ctx, 0);
    }

    // ---- verify_pbs (:388-489) on the LAST proof only ----
    vpbs_verify_inputs v{};
    v.log_n = cyc.log_n; v.rate_bits = 3; v.cap_height = 4;
    v.n_constants_sigmas = cyc.n_const_cols + cyc.n_routed; v.n_wires = cyc.n_wires; v.n_zs_partial_products = 20; v.n_quotient = 16;
    v.num_challenges = 2;
    v.constants_sigmas_cap = vk.data() + 4;
    for (int i = 0; i < 4; ++i) v.circuit_digest[i] = vk[i];
    v.n_constants = cyc.n_const_cols; v.n_routed = cyc.n_routed; v.quotient_degree_factor = 8;
    v.gates = cyc.gates.data(); v.n_gates = cyc.n_gates; v.num_selectors = cyc.num_selectors;
    std::vector<u64> b_caps(3 * 64), b_open(2 * ((size_t)v.n_constants_sigmas + 135 + 20 + 16 + 2)), b_fri(proof_words), b_pis(n_pi);
    REQUIRE(vpbs_step_proof_from_bytes(&v, bytes.data(), (size_t)n_bytes, b_caps.data(), b_open.data(), b_fri.data(), b_pis.data(), n_pi) == (long)n_pi,
            "from_bytes");
    v.public_inputs = b_pis.data(); v.n_public_inputs = n_pi;
    const double tv = now();
    const int ok = vpbs_verify_step(&v, b_caps.data(), b_open.data(), b_fri.data());
    const double verify_ms = 1e3 * (now() - tv);
    REQUIRE(ok == 1, "the final proof does not verify");
    REQUIRE(b_pis[kn] == steps, "counter");
    long decrypted = -1;
    if (steps == total) {   // the whole statement in one call of the library, then the decryption
        vpbs_verify_pbs_inputs vp{};
        vp.circuit = &v;
        vp.N = N; vp.K = K; vp.n_lwe = n_lwe; vp.ggsw_len = ggsw_len;
        vp.testv = testv.data(); vp.out_ct = b_pis.data() + kn + 1; vp.ct = ct.data(); vp.bsk = bsk.data(); vp.ksk = ksk.data();
        char why[256];
        REQUIRE(vpbs_verify_pbs(&vp, bytes.data(), (size_t)n_bytes, why, sizeof why) == 1, "verify_pbs: %s", why);
        std::vector<u64> m_bar(N);
        REQUIRE(vpbs_glwe_decrypt(ctx, log_N, K, s_to.data(), b_pis.data() + kn + 1, m_bar.data()) == 0, "decrypt");
        decrypted = (long)(((unsigned __int128)m_bar[0] * 2 + delta) / ((unsigned __int128)delta * 2)) % 4;   // round(m_bar / delta) mod 2 p
        REQUIRE(decrypted == (long)message, "the bootstrapped ciphertext decrypts to %ld, not %llu", decrypted, (unsigned long long)message);
    }
    const double seconds = t.seconds, t_late = t.late_witness_ms * steps / 1e3, t_prove = (t.late_rows_upload_ms + t.prove_step_ms) * steps / 1e3,
                 t_early = t.early_witness_ms * steps / 1e3;
    vpbs_ivc_free(ivc);
    std::printf("IVC chain: %u of %u step proofs of the cyclic circuit (%llu gate rows, degree 2^%u, %zu public inputs) in %.3f s "
                "(%.2f ms per step: late witness %.2f, late rows upload + prove %.2f; early phase on its thread %.2f); final proof %ld bytes, "
                "verified: %d in %.1f ms; decrypted %ld (message %llu)\n",
                steps, total, (unsigned long long)cyc.meta[5], cyc.log_n, n_pi, seconds, 1e3 * seconds / steps, 1e3 * t_late / steps,
                1e3 * t_prove / steps, 1e3 * t_early / steps, n_bytes, ok, verify_ms, decrypted, (unsigned long long)message);
    return 0;
}
