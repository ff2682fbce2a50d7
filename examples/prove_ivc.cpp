// One verifiable PBS as the reference produces it -- an IVC chain (/root/reference/src/vtfhe/ivc_based_vpbs.rs:159-386 `verified_pbs`,
// :388-489 `verify_pbs`) -- driven by a plain C++ host through the C ABI of include/vpbs_prover.h: no Python, no HIP runtime in this
// translation unit.  This is the shape of the Rust side's driver loop: the circuit arrives as data (the exported cyclic
// step circuit and its dummy circuit, tools/export_step_circuit.py --cyclic), and per step
//     PartialWitness = previous proof's words | its public inputs | condition | GGSW | mask | own verifier data | dummy verifier data
//     -> vpbs_witness_plan_run_early (everything that does not need the previous proof, on a second thread, ahead)
//     -> vpbs_device_upload_bg       (that matrix to the device, on a third thread, while the previous step is being proven)
//     -> vpbs_witness_plan_run_late  (the in-circuit verifier's rows, when the proof exists) -> vpbs_device_upload_rows (those rows only)
//     -> vpbs_prove_step (wires on the device) -> the proof feeds the next step.
// At the end: verify_pbs on the LAST proof only (byte round trip, vpbs_verify_step, test vector, counter, verifier data, chain hashes,
// decryption).  Keys, test vector and the LWE input: vpbs_keygen / vpbs_testv / vpbs_lwe_encrypt (seeded, the paper's noise).
//   build: g++ -O2 -std=c++17 -pthread -I include examples/prove_ivc.cpp -L verifiable-fhe-paper_amd -lvpbs_hip
//              -Wl,-rpath,$PWD/verifiable-fhe-paper_amd -o examples/prove_ivc
//   run  : python tools/export_step_circuit.py --cyclic /tmp/cyc.bin /tmp/dum.bin 8 2 4 5 6 13 && examples/prove_ivc /tmp/cyc.bin /tmp/dum.bin
//          examples/prove_ivc CYCLIC.bin DUMMY.bin [steps]        (steps < n + 2: a prefix of the chain)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
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
    std::vector<u64> sigma, cs_cap, vk;   // vk: circuit digest [4] then the constants/sigmas cap [16][4]
    u64* d_sigma = nullptr;
    vpbs_batch* cs = nullptr;
    vpbs_witness_plan* plan = nullptr;

    void load(const char* path, vpbs_ctx* ctx) {
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
        sigma.resize((size_t)n_routed * n);
        REQUIRE(vpbs_sigma_values(&circ, sigma.data()) == 0, "sigma values failed");
        // constants/sigmas commitment (once per circuit), verifier data, sigma values resident on the device
        std::vector<u64> csv(f_consts, f_consts + (size_t)n_const_cols * n);
        csv.insert(csv.end(), sigma.begin(), sigma.end());
        cs_cap.resize(64);
        REQUIRE(vpbs_commit_values(ctx, csv.data(), n_const_cols + n_routed, log_n, &cs, cs_cap.data()) == 0, "commit: %s", vpbs_last_error(ctx));
        std::vector<u64> dig_in(cs_cap);
        dig_in.push_back(log_n);
        vk.assign(4, 0);
        vpbs_hash_no_pad(dig_in.data(), dig_in.size(), vk.data());   // circuit digest: hash_no_pad(cap || degree bits)
        vk.insert(vk.end(), cs_cap.begin(), cs_cap.end());
        REQUIRE(vpbs_device_alloc(ctx, sigma.size(), &d_sigma) == 0 && vpbs_device_upload(ctx, d_sigma, sigma.data(), sigma.size()) == 0,
                "sigma upload: %s", vpbs_last_error(ctx));
        char err[256];
        REQUIRE(vpbs_witness_plan_create(&circ, preset_pos.data(), n_preset, &plan, err, sizeof err) == 0, "witness plan: %s", err);
    }

    void step_inputs(vpbs_step_inputs& in, const u64* wires_pinned, const u64* pis) const {
        in = vpbs_step_inputs{};
        in.log_n = log_n; in.n_wires = n_wires; in.n_zs_partial_products = 20; in.n_quotient = 16; in.num_challenges = 2;
        in.wires_values = wires_pinned;
        in.constants_sigmas = cs;
        for (int i = 0; i < 4; ++i) in.circuit_digest[i] = vk[i];
        in.public_inputs = pis; in.n_public_inputs = n_pi;
        in.forced_pow = VPBS_POW_ANY;
        in.sigmas_values = d_sigma; in.sigmas_on_device = 1;
        in.n_routed = n_routed; in.quotient_degree_factor = 8; in.n_constants = n_const_cols;
        in.gates = gates.data(); in.n_gates = n_gates; in.num_selectors = num_selectors;
    }
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
    cyc.load(argv[1], ctx);
    dum.load(argv[2], ctx);
    const unsigned N = (unsigned)cyc.meta[0], K = (unsigned)cyc.meta[1], ELL = (unsigned)cyc.meta[2], LOGB = (unsigned)cyc.meta[3],
                   n_lwe = (unsigned)cyc.meta[4];
    const size_t proof_words = cyc.meta[7], kn = (size_t)K * N, ggsw_len = (size_t)K * ELL * K * N, n_pi = cyc.n_pi;
    REQUIRE(cyc.meta[6] == 1 && dum.meta[6] == 2 && dum.n_preset == n_pi && n_pi == 2 * kn + 9 + 68, "not a cyclic / dummy pair");
    REQUIRE(cyc.n_preset == proof_words + n_pi + 1 + ggsw_len + 1 + 68 + 68, "unexpected PartialWitness layout");
    const unsigned total = n_lwe + 2;
    const unsigned steps = argc > 3 ? (unsigned)std::min<long>(total, std::atol(argv[3])) : total;
    unsigned log_N = 0;
    while ((1u << log_N) < N) ++log_N;

    // ---- main.rs:40-52 with seeded generators ----
    vpbs_keygen_params kp{log_N, K, ELL, LOGB, n_lwe, 0x5EED0728ull, 4.99027217501041e-8, 1.17021618159313e-5};
    std::vector<u64> s_lwe(n_lwe), s_glwe((size_t)(K - 1) * N), s_to((size_t)K * N), bsk((size_t)n_lwe * ggsw_len), ksk(ggsw_len), testv(N), ct(n_lwe + 1);
    REQUIRE(vpbs_keygen(ctx, &kp, s_lwe.data(), s_glwe.data(), s_to.data(), bsk.data(), ksk.data(), 0) == 0, "keygen: %s", vpbs_last_error(ctx));
    u64 delta = 0;
    REQUIRE(vpbs_testv(log_N, 2, testv.data(), &delta) == 0, "testv");
    const u64 message = 1;
    REQUIRE(vpbs_lwe_encrypt(&kp, s_lwe.data(), delta * message % P, 0, ct.data()) == 0, "lwe_encrypt");
    std::vector<u64> acc_init(kn, 0);
    std::copy(testv.begin(), testv.end(), acc_init.begin() + (kn - N));
    const std::vector<u64> zero_ggsw(ggsw_len, 0);
    auto ggsw_of = [&](unsigned s) { return s == 0 ? zero_ggsw.data() : (s <= n_lwe ? bsk.data() + (size_t)(s - 1) * ggsw_len : ksk.data()); };
    auto mask_of = [&](unsigned s) { return s == 0 ? ct[n_lwe] : (s <= n_lwe ? ct[s - 1] : (u64)0); };

    // ---- buffers: three pinned wire matrices and their device twins cycle through early thread -> uploader -> prover ----
    constexpr int NBUF = 3;
    const size_t wire_words = (size_t)cyc.n_wires * cyc.n;
    u64 *bufs[NBUF], *d_bufs[NBUF];
    for (auto& b : bufs) REQUIRE((b = static_cast<u64*>(vpbs_host_alloc(8 * wire_words))) != nullptr, "pinned allocation failed");
    for (auto& d : d_bufs) REQUIRE(vpbs_device_alloc(ctx, wire_words, &d) == 0, "device allocation: %s", vpbs_last_error(ctx));
    std::vector<uint8_t> late(cyc.n_preset, 0);
    std::fill(late.begin(), late.begin() + proof_words, 1);   // the previous proof's words arrive late
    char err[256];
    REQUIRE(vpbs_witness_plan_split(cyc.plan, late.data(), err, sizeof err) == 0, "split: %s", err);
    size_t late_rows[2];
    REQUIRE(vpbs_witness_plan_late_rows(cyc.plan, late_rows) == 0, "late rows");
    std::vector<u64> base_pis(acc_init);
    base_pis.resize(kn + 1 + kn + 8, 0);
    base_pis.insert(base_pis.end(), cyc.vk.begin(), cyc.vk.end());

    struct Ready {
        int buf;
        vpbs_witness_state* state;
        std::vector<u64> values, pis;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> free_bufs{0, 1, 2};
    std::deque<Ready> generated, ready;   // early thread -> uploader -> main
    std::atomic<bool> failed{false};
    double t_early = 0;
    auto values_of = [&](unsigned s, const std::vector<u64>& inner_pis) {
        std::vector<u64> v(proof_words, 0);
        v.insert(v.end(), inner_pis.begin(), inner_pis.end());
        v.push_back(s == 0 ? 0 : 1);                                    // condition: false only in the base step
        v.insert(v.end(), ggsw_of(s), ggsw_of(s) + ggsw_len);
        v.push_back(mask_of(s));
        v.insert(v.end(), cyc.vk.begin(), cyc.vk.end());
        v.insert(v.end(), dum.vk.begin(), dum.vk.end());
        return v;
    };
    std::thread early([&] {
        std::vector<u64> pis_prev(base_pis);
        char e2[256];
        for (unsigned s = 0; s < steps && !failed; ++s) {
            int b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !free_bufs.empty() || failed; });
                if (failed) return;
                b = free_bufs.front();
                free_bufs.pop_front();
            }
            const double t = now();
            Ready r{b, nullptr, values_of(s, pis_prev), {}};
            // the first pass through the three matrices fills them completely, later passes only rewrite the positions that carry values
            const auto run_early = s < NBUF ? vpbs_witness_plan_run_early : vpbs_witness_plan_run_early_recycled;
            if (run_early(cyc.plan, r.values.data(), 0, bufs[b], &r.state, e2, sizeof e2) != 0) {
                std::fprintf(stderr, "early witness phase of step %u: %s\n", s, e2);
                failed = true;
                cv.notify_all();
                return;
            }
            r.pis.resize(n_pi);
            for (size_t i = 0; i < n_pi; ++i) r.pis[i] = bufs[b][cyc.pi_pos[i]];   // public inputs never depend on the inner proof's words
            pis_prev = r.pis;
            t_early += now() - t;
            {
                std::lock_guard<std::mutex> lk(mu);
                generated.push_back(std::move(r));
            }
            cv.notify_all();
        }
    });
    std::thread uploader([&] {
        for (unsigned s = 0; s < steps; ++s) {
            Ready r;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !generated.empty() || failed; });
                if (failed) return;
                r = std::move(generated.front());
                generated.pop_front();
            }
            if (vpbs_device_upload_bg(ctx, d_bufs[r.buf], bufs[r.buf], wire_words) != 0) {
                std::fprintf(stderr, "upload of the early wires of step %u failed\n", s);
                failed = true;
                cv.notify_all();
                return;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                ready.push_back(std::move(r));
            }
            cv.notify_all();
        }
    });

    // ---- cyclic_base_proof (:292-299): a proof of the dummy circuit carrying the initial accumulator and the cyclic verifier data ----
    vpbs_step_inputs in;
    vpbs_step_sizes sz{};
    u64* base_wires = static_cast<u64*>(vpbs_host_alloc(8 * (size_t)dum.n_wires * dum.n));
    REQUIRE(base_wires && vpbs_witness_plan_run(dum.plan, base_pis.data(), 0, base_wires, err, sizeof err) == 0, "dummy witness: %s", err);
    dum.step_inputs(in, base_wires, base_pis.data());
    REQUIRE(vpbs_step_sizes_get(ctx, &in, &sz) == 0, "sizes");
    REQUIRE(3 * sz.cap_words + sz.openings_words + sz.fri_words == proof_words, "proof layout: %zu words expected", proof_words);
    std::vector<u64> proof(proof_words);
    u64 *caps = proof.data(), *openings = caps + 3 * sz.cap_words, *fri = openings + sz.openings_words;   // the flat order of the proof targets
    const double t0 = now();
    REQUIRE(vpbs_prove_step(ctx, &in, caps, openings, fri, nullptr, nullptr) == 0, "base proof: %s", vpbs_last_error(ctx));
    double t_late = 0, t_prove = 0;
    std::vector<u64> pis;
    const bool timing = std::getenv("VPBS_TIMING") != nullptr;   // HIP events around every kernel group of the step proofs
    if (timing) vpbs_timing_enable(ctx, 1);
    for (unsigned s = 0; s < steps; ++s) {
        Ready r;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || failed; });
            REQUIRE(!failed, "the early thread failed");
            r = std::move(ready.front());
            ready.pop_front();
        }
        double t = now();
        std::copy(proof.begin(), proof.end(), r.values.begin());
        REQUIRE(vpbs_witness_plan_run_late(cyc.plan, r.state, r.values.data(), bufs[r.buf], err, sizeof err) == 0,
                "late witness phase of step %u (the previous proof does not verify in circuit): %s", s, err);
        t_late += now() - t;
        t = now();
        pis = r.pis;
        REQUIRE(vpbs_device_upload_rows(ctx, d_bufs[r.buf], bufs[r.buf], cyc.n_wires, cyc.n, late_rows[0], late_rows[1]) == 0, "upload: %s",
                vpbs_last_error(ctx));
        cyc.step_inputs(in, d_bufs[r.buf], pis.data());
        in.inputs_on_device = 1;
        REQUIRE(vpbs_prove_step(ctx, &in, caps, openings, fri, nullptr, nullptr) == 0, "step %u: %s", s, vpbs_last_error(ctx));
        t_prove += now() - t;
        {
            std::lock_guard<std::mutex> lk(mu);
            free_bufs.push_back(r.buf);
        }
        cv.notify_all();
    }
    const double seconds = now() - t0;
    early.join();
    uploader.join();
    if (timing) {
        std::vector<char> report(1 << 14);
        vpbs_timing_report(ctx, report.data(), report.size());
        std::printf("device time of %u step proofs by kernel group: %s\n", steps, report.data());
        vpbs_timing_enable(ctx, 0);
    }

    // ---- verify_pbs (:388-489) on the LAST proof only ----
    std::vector<uint8_t> bytes(8 * (proof_words + n_pi) + 8192);
    const long n_bytes = vpbs_step_proof_to_bytes(ctx, &in, cyc.n_const_cols, caps, openings, fri, bytes.data(), bytes.size());
    REQUIRE(n_bytes > 0, "to_bytes");
    vpbs_verify_inputs v{};
    v.log_n = cyc.log_n; v.rate_bits = 3; v.cap_height = 4;
    v.n_constants_sigmas = cyc.n_const_cols + cyc.n_routed; v.n_wires = cyc.n_wires; v.n_zs_partial_products = 20; v.n_quotient = 16;
    v.num_challenges = 2;
    v.constants_sigmas_cap = cyc.cs_cap.data();
    for (int i = 0; i < 4; ++i) v.circuit_digest[i] = cyc.vk[i];
    v.n_constants = cyc.n_const_cols; v.n_routed = cyc.n_routed; v.quotient_degree_factor = 8;
    v.gates = cyc.gates.data(); v.n_gates = cyc.n_gates; v.num_selectors = cyc.num_selectors;
    std::vector<u64> b_caps(3 * sz.cap_words), b_open(sz.openings_words), b_fri(sz.fri_words), b_pis(n_pi);
    const long got_pis = vpbs_step_proof_from_bytes(&v, bytes.data(), (size_t)n_bytes, b_caps.data(), b_open.data(), b_fri.data(), b_pis.data(), n_pi);
    REQUIRE(got_pis == (long)n_pi, "from_bytes");
    v.public_inputs = b_pis.data(); v.n_public_inputs = n_pi;
    const double tv = now();
    const int ok = vpbs_verify_step(&v, b_caps.data(), b_open.data(), b_fri.data());
    const double verify_ms = 1e3 * (now() - tv);
    REQUIRE(ok == 1, "the final proof does not verify");
    REQUIRE(std::equal(acc_init.begin(), acc_init.end(), b_pis.begin()) && b_pis[kn] == steps, "test vector / counter");
    REQUIRE(std::equal(cyc.vk.begin(), cyc.vk.end(), b_pis.end() - 68), "check_cyclic_proof_verifier_data");
    std::vector<u64> items((size_t)steps * ggsw_len), masks(steps);
    for (unsigned s = 0; s < steps; ++s) {
        std::copy(ggsw_of(s), ggsw_of(s) + ggsw_len, items.begin() + (size_t)s * ggsw_len);
        masks[s] = mask_of(s);
    }
    REQUIRE(vpbs_hash_chain(items.data(), steps, ggsw_len, b_pis.data() + 2 * kn + 1, nullptr) == 1, "bootstrapping-key hash chain");
    REQUIRE(vpbs_hash_chain(masks.data(), steps, 1, b_pis.data() + 2 * kn + 5, nullptr) == 1, "LWE hash chain");
    if (steps == total) {   // the whole of verify_pbs in one call of the library
        vpbs_verify_pbs_inputs vp{};
        vp.circuit = &v;
        vp.N = N; vp.K = K; vp.n_lwe = n_lwe; vp.ggsw_len = ggsw_len;
        vp.testv = testv.data(); vp.out_ct = b_pis.data() + kn + 1; vp.ct = ct.data(); vp.bsk = bsk.data(); vp.ksk = ksk.data();
        char why[256];
        REQUIRE(vpbs_verify_pbs(&vp, bytes.data(), (size_t)n_bytes, why, sizeof why) == 1, "verify_pbs: %s", why);
    }
    long decrypted = -1;
    if (steps == total) {
        std::vector<u64> m_bar(N);
        REQUIRE(vpbs_glwe_decrypt(ctx, log_N, K, s_to.data(), b_pis.data() + kn + 1, m_bar.data()) == 0, "decrypt");
        decrypted = (long)(((unsigned __int128)m_bar[0] * 2 + delta) / ((unsigned __int128)delta * 2)) % 4;   // round(m_bar / delta) mod 2 p
        REQUIRE(decrypted == (long)message, "the bootstrapped ciphertext decrypts to %ld, not %llu", decrypted, (unsigned long long)message);
    }
    std::printf("IVC chain: %u of %u step proofs of the cyclic circuit (%llu gate rows, degree 2^%u, %zu public inputs) in %.3f s "
                "(%.2f ms per step: late witness %.2f, late rows upload + prove %.2f; early phase on its thread %.2f); final proof %ld bytes, "
                "verified: %d in %.1f ms; decrypted %ld (message %llu)\n",
                steps, total, (unsigned long long)cyc.meta[5], cyc.log_n, n_pi, seconds, 1e3 * seconds / steps, 1e3 * t_late / steps,
                1e3 * t_prove / steps, 1e3 * t_early / steps, n_bytes, ok, verify_ms, decrypted, (unsigned long long)message);
    return 0;
}
