// Pure C/C++ consumer of the C ABI (include/vpbs_prover.h): what a non-Python host -- the reference's Rust through
// `extern "C"`, or a C++ service -- does to prove one vPBS step.  No torch, no Python.
//   build: g++ -O2 -std=c++17 -I include examples/prove_step.cpp -L verifiable-fhe-paper_amd -lvpbs_hip \
//              -Wl,-rpath,$PWD/verifiable-fhe-paper_amd -o examples/prove_step
//   run  : examples/prove_step [log_n]      -> prints the three caps' first words, the pow witness and the proof size
// Inputs: the seeded synthetic step of SURVEY.md 8d (splitmix64 stream, values >= p rejected), identical to
// verifiable-fhe-paper_amd/synth.py, so the printed digest can be compared with the Python path.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vpbs_prover.h"

static const uint64_t P = 0xFFFFFFFF00000001ull;

static std::vector<uint64_t> field_elements(uint64_t seed, size_t count) {
    // synth.field_elements: splitmix64(seed) outputs with values >= p dropped; the Python version draws in blocks and
    // re-seeds each block with seed + drawn * golden, which is the same stream continued.
    std::vector<uint64_t> out;
    out.reserve(count);
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

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != 0) {                                                                     \
            std::fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, vpbs_last_error(ctx)); \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

int main(int argc, char** argv) {
    const unsigned log_n = argc > 1 ? (unsigned)std::atoi(argv[1]) : 12;
    const size_t n = (size_t)1 << log_n;
    const unsigned n_cs = 85, n_wires = 135, n_zs = 20, n_quot = 16, n_constants = 5, n_routed = 80;
    const uint64_t seed = 0x5EED0000ull;
    vpbs_ctx* ctx = nullptr;
    if (vpbs_ctx_create(0, 16, 3, 4, &ctx) != 0) {
        std::fprintf(stderr, "no MI355X device / context creation failed\n");
        return 2;
    }
    std::vector<uint64_t> wires = field_elements(seed, (size_t)n_wires * n);
    std::vector<uint64_t> quot = field_elements(seed + 2, (size_t)n_quot * n);
    std::vector<uint64_t> cs = field_elements(seed + 3, (size_t)n_cs * n);
    std::vector<uint64_t> pis = field_elements(0xABCD, 77);

    vpbs_batch* cs_batch = nullptr;
    std::vector<uint64_t> cs_cap(64);
    CHECK(vpbs_commit_values(ctx, cs.data(), n_cs, log_n, &cs_batch, cs_cap.data()));  // once per circuit

    vpbs_step_inputs in{};
    in.log_n = log_n;
    in.n_wires = n_wires;
    in.n_zs_partial_products = n_zs;
    in.n_quotient = n_quot;
    in.num_challenges = 2;
    in.inputs_on_device = 0;
    in.wires_values = wires.data();
    in.zs_pp_values = nullptr;  // Z / partial products computed on the device
    in.quotient_coeffs = quot.data();
    in.constants_sigmas = cs_batch;
    in.circuit_digest[0] = 11; in.circuit_digest[1] = 22; in.circuit_digest[2] = 33; in.circuit_digest[3] = 44;
    in.public_inputs = pis.data();
    in.n_public_inputs = pis.size();
    in.forced_pow = VPBS_POW_ANY;
    in.sigmas_values = cs.data() + (size_t)n_constants * n;
    in.n_routed = n_routed;
    in.quotient_degree_factor = 8;

    vpbs_step_sizes sz{};
    CHECK(vpbs_step_sizes_get(ctx, &in, &sz));
    std::vector<uint64_t> caps(3 * sz.cap_words), openings(sz.openings_words), fri(sz.fri_words);
    vpbs_challenger_state ch;
    CHECK(vpbs_prove_step(ctx, &in, caps.data(), openings.data(), fri.data(), &ch, nullptr));
    std::vector<uint8_t> bytes(8 * (caps.size() + openings.size() + fri.size() + pis.size() + 8) + 4096);
    const long nbytes = vpbs_step_proof_to_bytes(ctx, &in, n_constants, caps.data(), openings.data(), fri.data(), bytes.data(), bytes.size());
    if (nbytes < 0) return 1;
    // FNV-1a over the proof bytes: one number to compare across hosts
    uint64_t h = 0xcbf29ce484222325ull;
    for (long i = 0; i < nbytes; ++i) { h ^= bytes[i]; h *= 0x100000001b3ull; }
    std::printf("log_n=%u caps=%016llx,%016llx,%016llx pow=%llu proof_bytes=%ld fnv1a=%016llx\n", log_n, (unsigned long long)caps[0],
                (unsigned long long)caps[sz.cap_words], (unsigned long long)caps[2 * sz.cap_words], (unsigned long long)fri[sz.fri_words - 1],
                nbytes, (unsigned long long)h);
    vpbs_batch_free(cs_batch);
    vpbs_ctx_destroy(ctx);
    return 0;
}
