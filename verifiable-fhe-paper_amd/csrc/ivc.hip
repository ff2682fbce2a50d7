// One verifiable PBS as the reference produces it, as ONE call: the IVC chain of `verified_pbs`
// (/root/reference/src/vtfhe/ivc_based_vpbs.rs:159-386: build the cyclic step circuit, prove the base case, then n + 2 steps each verifying its
// predecessor in circuit) driven inside the library.  Host code only, written against the library's own C ABI (include/vpbs_prover.h) -- what
// examples/prove_ivc.cpp did by hand: nothing here reaches below that boundary.
//   create : (comm != NULL: coset-sharded over the GPUs of a node -- every rank calls with its own context and communicator, generates the
//            same witnesses, proves its cosets of every step and ends with the same proof)
//            constants/sigmas commitments and verifier data of the cyclic and the dummy circuit (CircuitBuilder::build's prover data),
//            compiled witness plans, the split of the cyclic plan (the previous proof's words are the late part), three pinned + three
//            device wire matrices
//   prove  : base proof of the dummy circuit (cyclic_base_proof, :292-299), then per step
//              thread E  vpbs_witness_plan_run_early   everything that does not need the previous proof; yields the step's public inputs
//              thread U  vpbs_device_upload_bg         that matrix to the device while the previous step is being proven
//              caller    vpbs_witness_plan_run_late -> vpbs_device_upload_rows -> vpbs_prove_step
//            and the last proof serialised (ProofWithPublicInputs::to_bytes)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <pthread.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vpbs_prover.h"

// The one thing this driver takes from the library beside the public header: vpbs_device_scatter's copy + kernel QUEUED on the context's
// stream without the wait (api.hip).  A host that writes its own loop over the C ABI calls vpbs_device_scatter and pays the wait.
namespace vpbs {
int device_scatter(vpbs_ctx* c, uint64_t* d_dst, const uint64_t* d_positions, const uint64_t* host_values, size_t count, uint64_t* d_stage, bool wait);
// ... and the late witness phase on the device stage by stage (witness_device.hip): queued as the previous proof's sections become final
unsigned witness_device_late_stages(const vpbs_witness_device* d);
int witness_device_run_late_stage(vpbs_witness_device* d, unsigned instance, unsigned stage, const uint64_t* preset_val, int wait);
}

namespace {
using u64 = uint64_t;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// the role of a thread in /proc/<pid>/task/*/comm: CPU time by role is how a host sizes a rank's CPU share (tools/prove_ivc.py VPBS_CPU_BY_ROLE)
void name_thread(const char* name) { (void)pthread_setname_np(pthread_self(), name); }
// CPU seconds the CALLING thread has used (VPBS_TRACE_IVC: what the proving thread burns while it waits for the device)
double thread_cpu() {
    timespec ts;
    return clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0 ? ts.tv_sec + 1e-9 * ts.tv_nsec : 0.0;
}

struct Side {   // one circuit on the context
    vpbs_ctx* ctx = nullptr;
    unsigned log_n = 0, n_wires = 0, n_routed = 0, n_const_cols = 0, num_selectors = 0;
    size_t n = 0;
    std::vector<vpbs_gate> gates;
    std::vector<uint32_t> pi_pos;   // cyclic circuit only (the dummy circuit's public inputs are its PartialWitness, never read back)
    size_t n_pi = 0, n_preset = 0;
    std::vector<u64> cs_cap, vk;   // vk: circuit digest [4] then the constants/sigmas cap
    u64* d_sigma = nullptr;
    u64* d_csv = nullptr;          // sharded commitment only: the constants / sigmas matrix on the device
    vpbs_batch* cs = nullptr;
    vpbs_witness_plan* plan = nullptr;
    const vpbs_comm* comm = nullptr;   // not null: every commitment and proof of this circuit is coset-sharded over comm->world GPUs

    int init(vpbs_ctx* c, const vpbs_ivc_circuit& d, unsigned cap_height, const vpbs_comm* cm, std::string& err) {
        ctx = c;
        comm = cm;
        const vpbs_circuit& k = *d.circuit;
        log_n = k.log_n; n_wires = k.n_wires; n_routed = k.n_routed; n_const_cols = k.n_constants_cols; num_selectors = k.num_selectors;
        n = (size_t)1 << log_n;
        gates.assign(k.gates, k.gates + k.n_gates);
        if (d.pi_pos) pi_pos.assign(d.pi_pos, d.pi_pos + d.n_pi);
        n_pi = d.n_pi;
        n_preset = d.n_preset;
        std::vector<u64> csv((size_t)(n_const_cols + n_routed) * n);
        std::memcpy(csv.data(), k.constants, 8 * (size_t)n_const_cols * n);
        u64* sigma = csv.data() + (size_t)n_const_cols * n;
        if (vpbs_sigma_values(&k, sigma) != 0) return err = "sigma polynomials: malformed copy constraints", VPBS_ERR_INVALID;
        const size_t cap_words = (size_t)4 << cap_height;
        cs_cap.resize(cap_words);
        int rc;
        if (!comm) {
            rc = vpbs_commit_values(ctx, csv.data(), n_const_cols + n_routed, log_n, &cs, cs_cap.data());
        } else {   // this rank's cosets; the cap is the concatenation of every rank's part (one all-gather of 32-byte hashes)
            if (comm->world == 0 || cap_words % comm->world != 0 || !comm->allgather) return err = "malformed communicator", VPBS_ERR_INVALID;
            rc = vpbs_device_alloc(ctx, csv.size(), &d_csv);
            if (rc == 0) rc = vpbs_device_upload(ctx, d_csv, csv.data(), csv.size());
            std::vector<u64> local(cap_words / comm->world);
            if (rc == 0) rc = vpbs_commit_sharded_dev(ctx, d_csv, 1, n_const_cols + n_routed, log_n, comm->rank, comm->world, &cs, local.data());
            // the all-gather carries every rank's status: a rank whose commitment failed still takes part, and all ranks return an error
            const int own = rc;
            rc = vpbs_comm_allgather_checked(comm, local.data(), local.size(), cs_cap.data(), own);
            if (rc != 0 && own == 0)
                return err = rc == VPBS_ERR_PEER ? "constants / sigmas commitment: another rank failed" : "all-gather of the cap failed", rc;
        }
        if (rc != 0) return err = std::string("constants / sigmas commitment: ") + vpbs_last_error(ctx), rc;
        vk.assign(4, 0);
        vpbs_compat compat;
        vpbs_ctx_get_compat(ctx, &compat);
        vpbs_circuit_digest(&compat, cs_cap.data(), cs_cap.size(), log_n, vk.data());   // CircuitBuilder::build's circuit_digest
        vk.insert(vk.end(), cs_cap.begin(), cs_cap.end());
        rc = vpbs_device_alloc(ctx, (size_t)n_routed * n, &d_sigma);
        if (rc == 0) rc = vpbs_device_upload(ctx, d_sigma, sigma, (size_t)n_routed * n);
        if (rc != 0) return err = std::string("sigma values to the device: ") + vpbs_last_error(ctx), rc;
        char e[256] = {0};
        rc = vpbs_witness_plan_create(&k, d.preset_pos, d.n_preset, &plan, e, sizeof e);
        if (rc != 0) return err = std::string("witness plan: ") + e, rc;
        return VPBS_OK;
    }
    void step_inputs(vpbs_step_inputs& in, const u64* wires, bool on_device, const u64* pis) const {
        in = vpbs_step_inputs{};
        in.log_n = log_n; in.n_wires = n_wires; in.n_zs_partial_products = 20; in.n_quotient = 16; in.num_challenges = 2;
        in.inputs_on_device = on_device ? 1 : 0;
        in.wires_values = wires;
        in.constants_sigmas = cs;
        for (int i = 0; i < 4; ++i) in.circuit_digest[i] = vk[i];
        in.public_inputs = pis; in.n_public_inputs = n_pi;
        in.forced_pow = VPBS_POW_ANY;
        in.sigmas_values = d_sigma; in.sigmas_on_device = 1;
        in.n_routed = n_routed; in.quotient_degree_factor = 8; in.n_constants = n_const_cols;
        in.gates = gates.data(); in.n_gates = (unsigned)gates.size(); in.num_selectors = num_selectors;
    }
    int prove(const vpbs_step_inputs& in, u64* caps, u64* openings, u64* fri) const {
        return comm ? vpbs_prove_step_sharded(ctx, &in, comm, caps, openings, fri, nullptr, nullptr)
                    : vpbs_prove_step(ctx, &in, caps, openings, fri, nullptr, nullptr);
    }
    void release() {
        if (plan) vpbs_witness_plan_free(plan);
        if (cs) vpbs_batch_free(cs);
        if (d_sigma) vpbs_device_free(ctx, d_sigma);
        if (d_csv) vpbs_device_free(ctx, d_csv);
        plan = nullptr; cs = nullptr; d_sigma = nullptr; d_csv = nullptr;
    }
};
// The late witness phase of step s + 1 in STAGES, started before the proof of step s is complete (the reference's loop
// ivc_based_vpbs.rs:323-353 hands a finished proof to the next step -- here the next step's in-circuit verifier starts on the sections of
// the proof as the prover finishes them).  The proof words are late presets of three stages (vpbs_witness_plan_split with stage numbers):
//   1            caps and openings        -> the in-circuit transcript up to the FRI challenges, the vanishing check at zeta, the reduced openings
//   2 .. 1 + R   the commit cap of FRI round r    -> the transcript absorbs it, the round's folding challenge
//   2 + R        final polynomial, proof-of-work witness -> the rest of the transcript, the query indices
//   3 + R        the query rounds         -> Merkle paths, folds: what is left on the critical path when the proof returns
// (R = 3 reduction rounds at degree 2^16.)  A worker thread runs all but the last on the next step's state (vpbs_witness_plan_run_late_stage) while the previous proof's FRI stage is
// still on the device; vpbs_step_inputs.on_section posts them.  The caller then drains the worker and runs vpbs_witness_plan_run_late_packed,
// which only has stage 3 left.
struct LateAhead {
    const vpbs_witness_plan* plan = nullptr;
    std::vector<std::vector<std::pair<size_t, size_t>>> ranges;   // per stage: [begin, end) word ranges of the flat proof
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    vpbs_witness_state* st = nullptr;   // the job: state and PartialWitness values of the NEXT step, the proof being written
    u64* values = nullptr;
    const u64* proof = nullptr;
    u64* packed = nullptr;              // the pinned buffer of the packed late wires: a stage leaves its share there at once
    unsigned posted = 0, done = 0;
    bool quit = false, failed = false;
    std::string err;
    double busy_s = 0;

    // stage of every proof word -> late mask for vpbs_witness_plan_split (the other presets: 0) and the ranges a stage copies
    void layout(size_t cap_words, size_t openings_words, size_t fri_words, unsigned n_rounds, size_t final_words, std::vector<uint8_t>& late) {
        const size_t head = 3 * cap_words + openings_words, fri_caps = (size_t)n_rounds * cap_words, tail = final_words + 1;
        ranges.assign(n_rounds + 3, {});
        ranges[0].push_back({0, head});                                                                  // section 1
        for (unsigned r = 0; r < n_rounds; ++r) ranges[1 + r].push_back({head + r * cap_words, head + (r + 1) * cap_words});   // sections 2 .. 1 + n_rounds
        ranges[n_rounds + 1].push_back({head + fri_words - tail, head + fri_words});                     // section 2 + n_rounds
        ranges[n_rounds + 2].push_back({head + fri_caps, head + fri_words - tail});                      // the return of the call
        for (size_t k = 0; k < ranges.size(); ++k)
            for (auto& r : ranges[k]) std::fill(late.begin() + r.first, late.begin() + r.second, (uint8_t)(k + 1));
    }
    void start(const vpbs_witness_plan* p) {
        plan = p;
        th = std::thread([this] {
            name_thread("vpbs-late-ahead");
            run();
        });
    }
    bool active() const { return st != nullptr; }
    void begin(vpbs_witness_state* state, u64* vals, const u64* proof_words) {   // proving thread, nothing posted yet
        std::lock_guard<std::mutex> lk(m);
        st = state; values = vals; proof = proof_words;
        posted = done = 0;
        failed = false;
        err.clear();
    }
    void post(unsigned stage) {   // proving thread (on_section): the words of stages <= stage are final
        {
            std::lock_guard<std::mutex> lk(m);
            posted = std::max(posted, std::min<unsigned>(stage, (unsigned)ranges.size() - 1));   // the last stage waits for the return
        }
        cv.notify_all();
    }
    // waits for the posted stages; afterwards the job is the caller's again (run_late for the rest).  false: a stage failed (msg)
    bool drain(std::string& msg) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done >= posted; });
        st = nullptr;
        if (failed) msg = err;
        return !failed;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    ~LateAhead() { stop(); }

  private:
    void run() {
        char e[256];
        for (;;) {
            unsigned stage;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || (st && done < posted); });
                if (quit) return;
                stage = done + 1;
            }
            const double t = now();
            bool ok = !failed;
            if (ok) {
                if (values != proof)
                    for (auto& r : ranges[stage - 1]) std::memcpy(values + r.first, proof + r.first, 8 * (r.second - r.first));
                e[0] = 0;
                ok = vpbs_witness_plan_run_late_stage(plan, st, stage, values, packed, e, sizeof e) == 0;
            }
            {
                std::lock_guard<std::mutex> lk(m);
                busy_s += now() - t;
                if (!ok && !failed) {
                    failed = true;
                    err = e;
                }
                done = stage;
            }
            cv.notify_all();
        }
    }
};
}  // namespace

struct vpbs_ivc {
    vpbs_ctx* ctx = nullptr;
    Side cyc, dum;
    unsigned N = 0, K = 0;
    size_t proof_words = 0, ggsw_len = 0, kn = 0, n_pi = 0, wire_words = 0;
    std::vector<u64> dummy_proof;   // the second proof slot: the dummy circuit's proof of all-zero public inputs (vpbs_ivc_create)
    bool staged = false;            // the cyclic plan is split into late stages by proof section (LateAhead)
    LateAhead ahead_layout;         // ranges only (the worker belongs to a prove_pbs call)
    size_t late_rows[2] = {0, 0};
    size_t late_count = 0;                       // wire positions the late witness phase writes (vpbs_witness_plan_late_count)
    u64 *late_vals = nullptr;                    // their values, packed, pinned (vpbs_witness_plan_run_late_packed)
    u64 *d_late_pos = nullptr, *d_late_stage = nullptr;   // device: the uint32 positions, the staging buffer of vpbs_device_scatter
    static constexpr int NBUF = 3;
    u64 *bufs[NBUF] = {nullptr, nullptr, nullptr}, *d_bufs[NBUF] = {nullptr, nullptr, nullptr}, *base_wires = nullptr;
    bool filled[NBUF] = {false, false, false};
    std::string err;
    vpbs_comm comm{};
    vpbs_ivc_step_fn step_fn = nullptr;
    void* step_user = nullptr;
    // early witness phases on the device (vpbs_ivc_set_device_witness): two early-only device objects, each on a context of its own (their
    // runs overlap with the prover's context and with each other's gathers), filled alternately with batches of `dw_batch` steps
    unsigned dw_batch = 0, ELL = 0, LOGB = 0;
    bool dw_late = false;   // the late phase on the device too (vpbs_witness_device_run_late): the host generates no witness at all
    vpbs_ctx* wctx[2] = {nullptr, nullptr};
    vpbs_witness_device* wdev[2] = {nullptr, nullptr};
    u64* dw_presets = nullptr;   // pinned [n_preset][dw_batch]
    size_t late_in_count = 0;
    void drop_device_witness() {
        for (int i = 0; i < 2; ++i) {
            if (wdev[i]) vpbs_witness_device_free(wdev[i]);
            if (wctx[i]) vpbs_ctx_destroy(wctx[i]);
            wdev[i] = nullptr;
            wctx[i] = nullptr;
        }
        if (dw_presets) vpbs_host_free(dw_presets);
        dw_presets = nullptr;
        dw_batch = 0;
    }
    ~vpbs_ivc() {
        for (auto b : bufs)
            if (b) vpbs_host_free(b);
        drop_device_witness();
        for (auto d : d_bufs)
            if (d) vpbs_device_free(ctx, d);
        if (base_wires) vpbs_host_free(base_wires);
        if (late_vals) vpbs_host_free(late_vals);
        if (d_late_pos) vpbs_device_free(ctx, d_late_pos);
        if (d_late_stage) vpbs_device_free(ctx, d_late_stage);
        cyc.release();
        dum.release();
    }
};

extern "C" {

int vpbs_ivc_create(vpbs_ctx* ctx, const vpbs_ivc_circuit* cyclic, const vpbs_ivc_circuit* dummy, unsigned N, unsigned K, size_t ggsw_len,
                    const vpbs_comm* comm, vpbs_ivc** out, char* err, size_t err_len) {
    auto say = [&](const std::string& m) {
        if (err && err_len) {
            std::strncpy(err, m.c_str(), err_len - 1);
            err[err_len - 1] = 0;
        }
    };
    say("");
    if (!ctx || !cyclic || !dummy || !out || !cyclic->circuit || !dummy->circuit || !cyclic->preset_pos || !dummy->preset_pos || !cyclic->pi_pos ||
        N == 0 || K == 0) {
        say("malformed arguments");
        return VPBS_ERR_INVALID;
    }
    // the proofs of the chain are proofs under CircuitConfig::standard_recursion_config (ivc_based_vpbs.rs:190): rate 1/8, cap height 4 -- the
    // in-circuit verifier of the cyclic circuit is built for that shape (16 cap entries in the verifier data and in every proof), so a
    // context of another shape cannot prove it: its commitments would carry another cap and every vk derived here would be wrong
    const unsigned cap_height = vpbs_ctx_cap_height(ctx);
    if (vpbs_ctx_rate_bits(ctx) != 3 || cap_height != 4) {
        say("the IVC chain needs a context with rate_bits = 3 and cap_height = 4 (standard_recursion_config)");
        return VPBS_ERR_INVALID;
    }
    const size_t kn = (size_t)K * N, cap_words = (size_t)4 << cap_height, n_pi = 2 * kn + 9 + 4 + cap_words;
    // the PartialWitness of a step, in the order the reference sets it (:314-330): previous proof | its public inputs | condition | GGSW |
    // mask | own verifier data | then what plonky2's DummyProofGenerator sets: dummy verifier data | the dummy circuit's proof | its public
    // inputs (the second proof slot of conditionally_verify_cyclic_proof_or_dummy); the dummy circuit's PartialWitness is its public inputs
    if (cyclic->n_pi != n_pi || dummy->n_preset != n_pi || (dummy->n_pi != 0 && dummy->n_pi != n_pi) ||
        cyclic->n_preset != 2 * (cyclic->proof_words + n_pi) + 1 + ggsw_len + 1 + 2 * (4 + cap_words)) {
        say("the circuits are not a cyclic step circuit and its dummy circuit for these parameters (public inputs / PartialWitness layout)");
        return VPBS_ERR_INVALID;
    }
    auto* v = new vpbs_ivc();
    v->ctx = ctx; v->N = N; v->K = K; v->ggsw_len = ggsw_len; v->kn = kn; v->n_pi = n_pi; v->proof_words = cyclic->proof_words;
    if (comm) v->comm = *comm;   // the callbacks and staging buffers stay the caller's; the struct itself is copied
    int rc = v->cyc.init(ctx, *cyclic, cap_height, comm ? &v->comm : nullptr, v->err);
    if (rc == 0) rc = v->dum.init(ctx, *dummy, cap_height, nullptr, v->err);   // the base proof is small: every rank proves it whole
    v->dum.n_pi = n_pi;   // the base proof carries the cyclic circuit's public-input layout whatever the caller wrote into dummy->n_pi
    if (rc == 0) {
        std::vector<uint8_t> late(cyclic->n_preset, 0);
        std::fill(late.begin(), late.begin() + cyclic->proof_words, 1);
        // the proof's sections as late STAGES (LateAhead); VPBS_IVC_LATE_STAGES=0: one late phase after the proof, as in rounds 2-3 (A-B runs)
        const char* ev = std::getenv("VPBS_IVC_LATE_STAGES");
        if (!(ev && std::atoi(ev) == 0)) {
            vpbs_fri_params fp;
            vpbs_fri_params_standard(v->cyc.log_n, &fp);
            unsigned final_bits = v->cyc.log_n;
            for (unsigned i = 0; i < fp.n_rounds; ++i) final_bits -= fp.arity_bits[i];
            vpbs_step_inputs shape;
            vpbs_step_sizes sz{};
            v->cyc.step_inputs(shape, nullptr, true, nullptr);
            const size_t fri_fixed = (size_t)fp.n_rounds * cap_words + ((size_t)2 << final_bits) + 1;
            if (vpbs_step_sizes_get(ctx, &shape, &sz) == 0 && sz.cap_words == cap_words &&
                3 * sz.cap_words + sz.openings_words + sz.fri_words == cyclic->proof_words && sz.fri_words > fri_fixed) {
                v->ahead_layout.layout(cap_words, sz.openings_words, sz.fri_words, fp.n_rounds, (size_t)2 << final_bits, late);
                v->staged = true;
            }
        }
        char e[256] = {0};
        rc = vpbs_witness_plan_split(v->cyc.plan, late.data(), e, sizeof e);
        if (rc != 0) v->err = std::string("split: ") + e;
        if (rc == 0) rc = vpbs_witness_plan_late_rows(v->cyc.plan, v->late_rows);
        if (rc == 0) v->late_count = vpbs_witness_plan_late_count(v->cyc.plan);
    }
    if (rc == 0) {
        v->wire_words = (size_t)v->cyc.n_wires * v->cyc.n;
        for (auto& b : v->bufs)
            if (!(b = static_cast<u64*>(vpbs_host_alloc(8 * v->wire_words)))) rc = VPBS_ERR_OOM;
        for (auto& d : v->d_bufs)
            if (rc == 0) rc = vpbs_device_alloc(ctx, v->wire_words, &d);
        if (!(v->base_wires = static_cast<u64*>(vpbs_host_alloc(8 * (size_t)v->dum.n_wires * v->dum.n)))) rc = VPBS_ERR_OOM;
        // the late phase's values travel packed (a few MB per step instead of the row range of all 135 columns) and are scattered on the device
        if (rc == 0 && v->late_count) {
            const size_t pos_words = (v->late_count + 1) / 2;
            if (!(v->late_vals = static_cast<u64*>(vpbs_host_alloc(8 * std::max(v->late_count, pos_words))))) rc = VPBS_ERR_OOM;
            if (rc == 0) rc = vpbs_device_alloc(ctx, pos_words, &v->d_late_pos);
            if (rc == 0) rc = vpbs_device_alloc(ctx, v->late_count, &v->d_late_stage);
            if (rc == 0) rc = vpbs_witness_plan_late_positions(v->cyc.plan, reinterpret_cast<uint32_t*>(v->late_vals));
            if (rc == 0) rc = vpbs_device_upload(ctx, v->d_late_pos, v->late_vals, pos_words);
        }
        if (rc != 0) v->err = "wire matrices: out of (pinned or device) memory";
    }
    if (rc == 0) {
        // dummy_proof_and_vk (recursion/dummy_circuit.rs): the dummy circuit's proof of all-zero public inputs -- the content of the second
        // proof slot in every step of every chain, so it is proven once here
        char e[256] = {0};
        const std::vector<u64> zero_pis(n_pi, 0);
        vpbs_step_inputs in;
        vpbs_step_sizes sz{};
        if (vpbs_witness_plan_run(v->dum.plan, zero_pis.data(), 0, v->base_wires, e, sizeof e) != 0) {
            v->err = std::string("dummy witness: ") + e;
            rc = VPBS_ERR_INVALID;
        }
        if (rc == 0) {
            v->dum.step_inputs(in, v->base_wires, false, zero_pis.data());
            if (vpbs_step_sizes_get(ctx, &in, &sz) != 0 || 3 * sz.cap_words + sz.openings_words + sz.fri_words != v->proof_words) {
                v->err = "the proof of this shape does not have the number of words the cyclic circuit expects";
                rc = VPBS_ERR_INVALID;
            }
        }
        if (rc == 0) {
            v->dummy_proof.assign(v->proof_words, 0);
            u64 *caps = v->dummy_proof.data(), *openings = caps + 3 * sz.cap_words, *fri = openings + sz.openings_words;
            rc = v->dum.prove(in, caps, openings, fri);
            if (rc != 0) v->err = std::string("dummy proof: ") + vpbs_last_error(ctx);
        }
    }
    if (rc != 0) {
        say(v->err);
        delete v;
        return rc;
    }
    *out = v;
    return VPBS_OK;
}

void vpbs_ivc_free(vpbs_ivc* v) { delete v; }

int vpbs_ivc_set_step_callback(vpbs_ivc* v, vpbs_ivc_step_fn fn, void* user) {
    if (!v) return VPBS_ERR_INVALID;
    v->step_fn = fn;
    v->step_user = user;
    return VPBS_OK;
}

const char* vpbs_ivc_last_error(const vpbs_ivc* v) { return v ? v->err.c_str() : ""; }

int vpbs_ivc_verifier_data(const vpbs_ivc* v, uint64_t* cyclic_vk, uint64_t* dummy_vk) {
    if (!v) return VPBS_ERR_INVALID;
    if (cyclic_vk) std::memcpy(cyclic_vk, v->cyc.vk.data(), 8 * v->cyc.vk.size());
    if (dummy_vk) std::memcpy(dummy_vk, v->dum.vk.data(), 8 * v->dum.vk.size());
    return VPBS_OK;
}

int vpbs_ivc_set_device_witness(vpbs_ivc* v, unsigned ELL, unsigned LOGB, unsigned batch, int late_on_device) {
    if (!v) return VPBS_ERR_INVALID;
    v->drop_device_witness();
    v->dw_late = false;
    v->err.clear();
    if (batch == 0) return VPBS_OK;
    if (ELL == 0 || LOGB == 0 || v->ggsw_len != (size_t)v->K * ELL * v->K * v->N) {
        v->err = "device witness: ELL / LOGB do not fit the circuit's GGSW length";
        return VPBS_ERR_INVALID;
    }
    int rc = VPBS_OK;
    for (int i = 0; i < 2 && rc == 0; ++i) {
        rc = vpbs_ctx_create(vpbs_ctx_device(v->ctx), v->cyc.log_n, vpbs_ctx_rate_bits(v->ctx), vpbs_ctx_cap_height(v->ctx), &v->wctx[i]);
        if (rc == 0) rc = vpbs_witness_device_create_early(v->wctx[i], v->cyc.plan, batch, &v->wdev[i]);
        if (rc != 0) v->err = std::string("device witness: ") + (v->wctx[i] ? vpbs_last_error(v->wctx[i]) : "no context");
        if (rc == 0 && late_on_device && !vpbs_witness_device_has_late(v->wdev[i])) {
            v->err = "device witness: late_on_device was requested, but the late phase of this circuit has no device schedule (a late generator "
                     "without a device form); use late_on_device = 0";
            rc = VPBS_ERR_INVALID;
        }
    }
    if (rc == 0 && !(v->dw_presets = static_cast<u64*>(vpbs_host_alloc(8 * v->cyc.n_preset * (size_t)batch)))) {
        v->err = "device witness: out of pinned memory";
        rc = VPBS_ERR_OOM;
    }
    if (rc != 0) {
        v->drop_device_witness();
        return rc;
    }
    v->late_in_count = vpbs_witness_plan_late_input_count(v->cyc.plan);
    v->ELL = ELL; v->LOGB = LOGB; v->dw_batch = batch;
    v->dw_late = late_on_device != 0;
    return VPBS_OK;
}

// The chain with the early witness phases on the device.  What a step's early phase needs of its predecessor are the predecessor's PUBLIC
// INPUTS, and those are known without proving anything: the accumulators from the native chain (vpbs_pbs_accumulator_chain), the two chain
// hashes from the native sponge (one host thread walks them ahead of the batches), counter and verifier data.  So the early phases of
// `dw_batch` consecutive steps run on the device at once (thread B), a stager thread gathers an instance's wires into the device matrix
// the prover will read and fetches the early values the late phase needs (thread S), and the caller is left with the late phase (the
// in-circuit verifier's rows), the scatter of its values and the proof.
static long prove_pbs_device_witness(vpbs_ivc* v, const uint64_t* testv, const uint64_t* ct, const uint64_t* bsk, const uint64_t* ksk, unsigned n_lwe,
                                     unsigned steps, uint8_t* proof_out, size_t capacity, vpbs_ivc_timing* timing, char* err, size_t err_len) {
    auto say = [&](const std::string& m) {
        if (err && err_len) {
            std::strncpy(err, m.c_str(), err_len - 1);
            err[err_len - 1] = 0;
        }
    };
    Side &cyc = v->cyc, &dum = v->dum;
    vpbs_ctx* ctx = v->ctx;
    const size_t kn = v->kn, n_pi = v->n_pi, proof_words = v->proof_words, ggsw_len = v->ggsw_len, n_preset = cyc.n_preset;
    const unsigned B = v->dw_batch, total = n_lwe + 2;
    const std::vector<u64> zero_ggsw(ggsw_len, 0);
    auto ggsw_of = [&](unsigned s) { return s == 0 ? zero_ggsw.data() : (s <= n_lwe ? bsk + (size_t)(s - 1) * ggsw_len : ksk); };
    auto mask_of = [&](unsigned s) { return s == 0 ? ct[n_lwe] : (s <= n_lwe ? ct[s - 1] : (u64)0); };
    const double t0 = now();
    // ---- the chain's public inputs, natively ----
    std::vector<u64> acc_init(kn, 0), accs((size_t)total * kn);
    std::memcpy(acc_init.data() + kn - v->N, testv, 8 * (size_t)v->N);
    unsigned log_N = 0;
    while ((1u << log_N) < v->N) ++log_N;
    const vpbs_tfhe_params tp{log_N, v->K, v->ELL, v->LOGB};
    if (vpbs_pbs_accumulator_chain(ctx, &tp, n_lwe, acc_init.data(), ct, bsk, ksk, accs.data()) != 0) {
        say(std::string("native accumulator chain: ") + vpbs_last_error(ctx));
        return VPBS_ERR_INVALID;
    }
    // pis[s + 1] = public inputs of step s (pis[0]: of the base proof): acc_init | counter | accumulator | key hash | LWE hash | verifier data
    std::vector<u64> pis((size_t)(steps + 1) * n_pi, 0);
    for (unsigned s = 0; s <= steps; ++s) {
        u64* q = pis.data() + (size_t)s * n_pi;
        std::memcpy(q, acc_init.data(), 8 * kn);
        q[kn] = s;
        if (s) std::memcpy(q + kn + 1, accs.data() + (size_t)(s - 1) * kn, 8 * kn);
        std::memcpy(q + n_pi - cyc.vk.size(), cyc.vk.data(), 8 * cyc.vk.size());
    }
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<bool> failed{false};
    std::string thread_err;
    auto fail = [&](const std::string& m) {
        {   // the flag changes under the lock: a waiter that has just evaluated its predicate cannot miss the notification
            std::lock_guard<std::mutex> lk(mu);
            if (thread_err.empty()) thread_err = m;
            failed = true;
        }
        cv.notify_all();
    };
    unsigned hashed = 0;        // steps whose chain hashes are in pis (guarded by mu)
    unsigned batches_run = 0;   // batches whose early phases are on the device
    unsigned staged = 0;        // steps gathered and read back
    unsigned consumed = 0;      // steps the caller has finished with (their device matrix is free again)
    double t_early = 0;
    // thread H: the two hash chains of verify_hash_output (:64-78), h_s = hash_no_pad(h_{s-1} || item_s), one permutation after the other
    std::thread hasher([&] {
        name_thread("vpbs-hash");
        std::vector<u64> in2(5);
        u64 hb[4] = {0, 0, 0, 0}, hl[4] = {0, 0, 0, 0};
        // the key chain in segments of eight links: the chains of one process walk theirs side by side (vpbs_hash_chain_links)
        constexpr unsigned SEG = 8;
        const u64* items[SEG];
        u64 links[4 * SEG];
        for (unsigned s0 = 0; s0 < steps && !failed; s0 += SEG) {
            {   // three batches ahead of the chain, no further: the batcher needs two, and the chains' hashing is spread over their whole
                // length instead of a burst of 1.9 s of CPU per chain at the start (which a timed window further on would not see)
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || s0 < consumed + 3 * B; });
                if (failed) return;
            }
            const unsigned cnt = std::min(SEG, steps - s0);
            for (unsigned i = 0; i < cnt; ++i) items[i] = ggsw_of(s0 + i);
            if (vpbs_hash_chain_links(hb, items, cnt, ggsw_len, links) != 0) return fail("hash chain of the bootstrapping key: malformed arguments");
            std::memcpy(hb, links + 4 * (cnt - 1), 32);
            for (unsigned i = 0; i < cnt; ++i) {
                const unsigned s = s0 + i;
                std::memcpy(in2.data(), hl, 32);
                in2[4] = mask_of(s);
                vpbs_hash_no_pad(in2.data(), 5, hl);
                u64* q = pis.data() + (size_t)(s + 1) * n_pi + 2 * kn + 1;
                std::memcpy(q, links + 4 * i, 32);
                std::memcpy(q + 4, hl, 32);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                hashed = s0 + cnt;
            }
            cv.notify_all();
        }
    });
    // thread B: batch b = steps [b B, (b + 1) B) on device object b & 1, once their public inputs exist and the object's previous batch is consumed
    const unsigned n_batches = (steps + B - 1) / B;
    std::thread batcher([&] {
        name_thread("vpbs-batcher");
        for (unsigned b = 0; b < n_batches && !failed; ++b) {
            const unsigned first = b * B, cnt = std::min(B, steps - first);
            {
                std::unique_lock<std::mutex> lk(mu);
                // step s needs the public inputs of step s - 1: hashed >= first + cnt - 1; the object held batch b - 2
                cv.wait(lk, [&] { return failed || (hashed + 1 >= first + cnt && (b < 2 || consumed >= (b - 1) * B)); });
                if (failed) return;
            }
            const double t = now();
            u64* m = v->dw_presets;   // [n_preset][cnt]: previous proof (late: ignored) | its public inputs | condition | GGSW | mask | vks | dummy proof | its pis
            auto row = [&](size_t r) { return m + r * cnt; };
            for (size_t r = 0; r < proof_words; ++r) std::memset(row(r), 0, 8 * cnt);
            for (unsigned i = 0; i < cnt; ++i) {
                const unsigned s = first + i;
                const u64* q = pis.data() + (size_t)s * n_pi;
                size_t r = proof_words;
                for (size_t k = 0; k < n_pi; ++k) row(r++)[i] = q[k];
                row(r++)[i] = s == 0 ? 0 : 1;
                const u64* g = ggsw_of(s);
                for (size_t k = 0; k < ggsw_len; ++k) row(r++)[i] = g[k];
                row(r++)[i] = mask_of(s);
                for (u64 x : cyc.vk) row(r++)[i] = x;
                for (u64 x : dum.vk) row(r++)[i] = x;
                for (u64 x : v->dummy_proof) row(r++)[i] = x;
                for (size_t k = 0; k < n_pi; ++k) row(r++)[i] = 0;
            }
            if (vpbs_witness_device_run(v->wdev[b & 1], m, cnt) != 0)
                return fail("early witness phases of steps " + std::to_string(first) + ".. on the device: " + vpbs_last_error(v->wctx[b & 1]));
            t_early += now() - t;
            {
                std::lock_guard<std::mutex> lk(mu);
                batches_run = b + 1;
            }
            cv.notify_all();
        }
    });
    // thread S: per step, the instance's wires into one of the device matrices, the late phase's inputs to the host and the late phase's
    // state seeded with them (allocation and first touch of its pages happen here, ahead of the chain)
    std::vector<std::vector<u64>> late_in(vpbs_ivc::NBUF, std::vector<u64>(v->late_in_count));
    struct States {   // owned here until the caller takes one; whatever is left at the end is freed
        vpbs_witness_state* st[vpbs_ivc::NBUF] = {nullptr, nullptr, nullptr};
        ~States() {
            for (auto* x : st)
                if (x) vpbs_witness_state_free(x);
        }
    } states;
    std::thread stager([&] {
        name_thread("vpbs-stager");
        if (v->dw_late) return;   // the caller runs the late phase on the device object itself and gathers afterwards
        for (unsigned s = 0; s < steps && !failed; ++s) {
            const unsigned b = s / B, k = s % vpbs_ivc::NBUF;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || (batches_run > b && s < consumed + vpbs_ivc::NBUF); });
                if (failed) return;
            }
            if (vpbs_witness_device_wires(v->wdev[b & 1], s % B, v->d_bufs[k]) != 0 ||
                vpbs_witness_device_read_late_inputs(v->wdev[b & 1], s % B, late_in[k].data()) != 0)
                return fail("gathering the early wires of step " + std::to_string(s) + ": " + vpbs_last_error(v->wctx[b & 1]));
            vpbs_witness_state* st = nullptr;
            if (vpbs_witness_state_from_late_inputs(cyc.plan, late_in[k].data(), &st) != 0)
                return fail("late witness phase of step " + std::to_string(s) + ": the early values read back from the device are malformed");
            {
                std::lock_guard<std::mutex> lk(mu);
                states.st[k] = st;
                staged = s + 1;
            }
            cv.notify_all();
        }
    });
    auto stop = [&](const std::string& m, int rc) {
        fail(m);
        hasher.join();
        batcher.join();
        stager.join();
        std::string first;
        {
            std::lock_guard<std::mutex> lk(mu);
            first = thread_err;
        }
        say(first);
        return (long)rc;
    };
    // cyclic_base_proof (:292-299)
    char e[256] = {0};
    vpbs_step_inputs in;
    vpbs_step_sizes sz{};
    if (vpbs_witness_plan_run(dum.plan, pis.data(), 0, v->base_wires, e, sizeof e) != 0) return stop(std::string("dummy witness: ") + e, VPBS_ERR_INVALID);
    dum.step_inputs(in, v->base_wires, false, pis.data());
    if (vpbs_step_sizes_get(ctx, &in, &sz) != 0 || 3 * sz.cap_words + sz.openings_words + sz.fri_words != proof_words)
        return stop("the proof of this shape does not have the number of words the cyclic circuit expects", VPBS_ERR_INVALID);
    std::vector<u64> values(n_preset, 0);   // run_late reads the late presets only: the previous proof's words, at the front
    u64 *caps = values.data(), *openings = caps + 3 * sz.cap_words, *fri = openings + sz.openings_words;
    int rc = dum.prove(in, caps, openings, fri);
    if (rc != 0) return stop(std::string("base proof: ") + vpbs_last_error(ctx), rc);
    const double t_base = now() - t0;
    if (v->step_fn) v->step_fn(v->step_user, 0);
    double t_late = 0, t_rows = 0, t_prove = 0, t_wait_staged = 0, t_wait_hashed = 0;
    double c_late = 0, c_rows = 0, c_prove = 0;   // CPU time of THIS thread in the same sections (VPBS_TRACE_IVC)
    // late stages 1 and 2 of the next step while this step's FRI stage runs (LateAhead); here the prover writes the proof straight into
    // `values`, the array the late phase reads its presets from
    vpbs_witness_state* next_state = nullptr;   // taken from `states` by the hook; declared before the worker (joined first)
    struct NextGuard {
        vpbs_witness_state*& p;
        ~NextGuard() {
            if (p) vpbs_witness_state_free(p);
        }
    } next_guard{next_state};
    LateAhead ahead;
    if (v->staged && !v->dw_late) {
        ahead.ranges = v->ahead_layout.ranges;
        ahead.packed = v->late_vals;
        ahead.start(cyc.plan);
    }
    struct Hook {
        LateAhead* ahead;
        std::mutex* mu;
        const unsigned* staged;
        States* states;
        vpbs_witness_state** next_state;
        u64* values;
        unsigned step = 0;   // the step being proven
        static void section(void* user, int sec) {
            auto* h = static_cast<Hook*>(user);
            if (!*h->next_state) {
                std::lock_guard<std::mutex> lk(*h->mu);
                if (*h->staged <= h->step + 1) return;   // the next step is not staged yet: its late phase runs whole, after the proof
                const unsigned k = (h->step + 1) % vpbs_ivc::NBUF;
                *h->next_state = h->states->st[k];
                h->states->st[k] = nullptr;
                h->ahead->begin(*h->next_state, h->values, h->values);
            }
            h->ahead->post((unsigned)sec);
        }
    } hook{&ahead, &mu, &staged, &states, &next_state, values.data()};
    // The same idea with the late phase on the DEVICE (round 6): the stages of the next step's late phase are queued on its witness object's
    // stream as the sections of the proof in progress become final (the prover writes them straight into `values`); when the proof returns only
    // the last stage -- the query rounds: 33 of the 162 dependency levels -- is left to queue and wait for.
    struct DevHook {
        vpbs_ivc* v;
        std::mutex* mu;
        const unsigned* batches_run;
        unsigned B;
        const u64* values;
        unsigned step = 0, queued = 0;   // the step being proven; stages of step + 1 queued so far
        static void section(void* user, int sec) {
            auto* h = static_cast<DevHook*>(user);
            const unsigned next = h->step + 1;
            {
                std::lock_guard<std::mutex> lk(*h->mu);
                if (*h->batches_run <= next / h->B) return;   // the next step's early batch has not run yet: its late phase runs whole, after the proof
            }
            vpbs_witness_device* dev = h->v->wdev[(next / h->B) & 1];
            const unsigned n = vpbs::witness_device_late_stages(dev);
            while (h->queued < (unsigned)sec && h->queued + 1 < n) {   // the last stage waits for the return
                if (vpbs::witness_device_run_late_stage(dev, next % h->B, h->queued + 1, h->values, 0) != 0) return;   // run_late reports it
                ++h->queued;
            }
        }
    } dev_hook{v, &mu, &batches_run, B, values.data()};
    // a rank of a SHARDED chain that cannot start a step's proof takes part in the step's collectives on the failing side (see the host pipeline)
    auto stop_step = [&](const std::string& m, int why) {
        if (cyc.comm) {
            vpbs_step_inputs shape;
            cyc.step_inputs(shape, nullptr, true, nullptr);
            (void)vpbs_prove_step_sharded_fail(ctx, &shape, cyc.comm, why);
        }
        return stop(m, why);
    };
    for (unsigned s = 0; s < steps; ++s) {
        const unsigned k = s % vpbs_ivc::NBUF;
        {
            const double tw = now();
            std::unique_lock<std::mutex> lk(mu);
            const unsigned b = s / B;
            cv.wait(lk, [&] { return failed || (v->dw_late ? batches_run > b : staged > s); });
            const double tw2 = now();
            t_wait_staged += tw2 - tw;
            cv.wait(lk, [&] { return failed || hashed > s; });   // its own public inputs are complete
            t_wait_hashed += now() - tw2;
            if (failed) {
                lk.unlock();
                return stop_step("", VPBS_ERR_INVALID);
            }
        }
        double t = now();
        double c0 = thread_cpu();
        if (v->dw_late) {
            // the late phase on the device object that holds the step's early values, then ALL its wires into the prover's matrix
            vpbs_witness_device* dev = v->wdev[(s / B) & 1];
            rc = vpbs_witness_device_run_late(dev, s % B, values.data());
            if (rc != 0)
                return stop_step("late witness phase of step " + std::to_string(s) + " (the previous proof does not verify in circuit): " +
                            vpbs_last_error(v->wctx[(s / B) & 1]), rc);
            t_late += now() - t;
            t = now();
            rc = vpbs_witness_device_wires(dev, s % B, v->d_bufs[k]);
            if (rc != 0) return stop_step(std::string("gathering the wires: ") + vpbs_last_error(v->wctx[(s / B) & 1]), rc);
            t_rows += now() - t;
        } else {
            vpbs_witness_state* st = next_state;
            next_state = nullptr;
            if (st) {   // stages ran ahead on it: wait for the last of them
                std::string msg;
                if (!ahead.drain(msg)) {
                    vpbs_witness_state_free(st);
                    return stop_step("late witness phase of step " + std::to_string(s) + " (the previous proof does not verify in circuit): " + msg, VPBS_ERR_INVALID);
                }
            } else {
                std::lock_guard<std::mutex> lk(mu);
                st = states.st[k];
                states.st[k] = nullptr;
            }
            rc = vpbs_witness_plan_run_late_packed(cyc.plan, st, values.data(), v->late_vals, e, sizeof e);
            if (rc != 0) return stop_step("late witness phase of step " + std::to_string(s) + " (the previous proof does not verify in circuit): " + e, rc);
            t_late += now() - t;
            t = now();
            c_late += thread_cpu() - c0;
            c0 = thread_cpu();
            // queued, not waited for: the proof is ordered behind it on the context's stream, and the packed buffer is next written a whole
            // proof (several waits on this stream) later -- by the late stages of the NEXT step, which start at this proof's first section
            rc = vpbs::device_scatter(ctx, v->d_bufs[k], v->d_late_pos, v->late_vals, v->late_count, v->d_late_stage, false);
            if (rc != 0) return stop_step(std::string("upload of the late wires: ") + vpbs_last_error(ctx), rc);
            t_rows += now() - t;
            c_rows += thread_cpu() - c0;
        }
        t = now();
        c0 = thread_cpu();
        cyc.step_inputs(in, v->d_bufs[k], true, pis.data() + (size_t)(s + 1) * n_pi);
        if (v->staged && !v->dw_late && s + 1 < steps) {
            hook.step = s;
            in.on_section = &Hook::section;
            in.on_section_user = &hook;
        }
        static const bool dev_ahead = !std::getenv("VPBS_DEVICE_LATE_AHEAD") || std::atoi(std::getenv("VPBS_DEVICE_LATE_AHEAD")) != 0;   // A-B switch
        if (v->staged && v->dw_late && dev_ahead && s + 1 < steps) {
            dev_hook.step = s;
            dev_hook.queued = 0;
            in.on_section = &DevHook::section;
            in.on_section_user = &dev_hook;
        }
        rc = cyc.prove(in, caps, openings, fri);
        if (rc != 0) {
            if (ahead.active()) {   // the worker may be inside a stage of the next step's state
                std::string ignored;
                (void)ahead.drain(ignored);
            }
            return stop("step " + std::to_string(s) + ": " + vpbs_last_error(ctx), rc);
        }
        t_prove += now() - t;
        c_prove += thread_cpu() - c0;
        {
            std::lock_guard<std::mutex> lk(mu);
            consumed = s + 1;
        }
        cv.notify_all();
        if (v->step_fn) v->step_fn(v->step_user, s + 1);
    }
    hasher.join();
    batcher.join();
    stager.join();
    const double seconds = now() - t0;
    if (std::getenv("VPBS_TRACE_IVC"))
        std::fprintf(stderr, "[ivc device witness] per step: waited %.2f ms for the staged wires, %.2f ms for the hash chain; late %.2f, scatter %.2f, "
                     "prove %.2f, device batch run %.2f ms; CPU time of the proving thread: late %.2f, scatter %.2f, prove %.2f ms (blocking waits: %d)\n",
                     1e3 * t_wait_staged / steps, 1e3 * t_wait_hashed / steps, 1e3 * t_late / steps,
                     1e3 * t_rows / steps, 1e3 * t_prove / steps, 1e3 * t_early / steps, 1e3 * c_late / steps, 1e3 * c_rows / steps, 1e3 * c_prove / steps,
                     vpbs_host_blocking_sync());
    const long n_bytes = vpbs_step_proof_to_bytes(ctx, &in, cyc.n_const_cols, caps, openings, fri, proof_out, capacity);
    if (n_bytes <= 0) {
        say("the output buffer is too small for the proof");
        return VPBS_ERR_INVALID;
    }
    if (timing) {
        timing->seconds = seconds;
        timing->steps = steps;
        timing->base_proof_ms = 1e3 * t_base;
        timing->late_witness_ms = 1e3 * t_late / steps;
        timing->late_rows_upload_ms = 1e3 * t_rows / steps;
        timing->prove_step_ms = 1e3 * t_prove / steps;
        timing->early_witness_ms = 1e3 * t_early / steps;
        timing->late_ahead_ms = 1e3 * ahead.busy_s / steps;
    }
    return n_bytes;
}

long vpbs_ivc_prove_pbs(vpbs_ivc* v, const uint64_t* testv, const uint64_t* ct, const uint64_t* bsk, const uint64_t* ksk, unsigned n_lwe,
                        unsigned steps, uint8_t* proof_out, size_t capacity, vpbs_ivc_timing* timing, char* err, size_t err_len) {
    auto say = [&](const std::string& m) {
        if (err && err_len) {
            std::strncpy(err, m.c_str(), err_len - 1);
            err[err_len - 1] = 0;
        }
    };
    say("");
    if (!v || !testv || !ct || !ksk || (n_lwe && !bsk) || !proof_out) {
        say("malformed arguments");
        return VPBS_ERR_INVALID;
    }
    const unsigned total = n_lwe + 2;
    if (steps == 0 || steps > total) steps = total;
    if (v->dw_batch) return prove_pbs_device_witness(v, testv, ct, bsk, ksk, n_lwe, steps, proof_out, capacity, timing, err, err_len);
    Side &cyc = v->cyc, &dum = v->dum;
    vpbs_ctx* ctx = v->ctx;
    const size_t kn = v->kn, n_pi = v->n_pi, proof_words = v->proof_words, ggsw_len = v->ggsw_len;
    const std::vector<u64> zero_ggsw(ggsw_len, 0);
    auto ggsw_of = [&](unsigned s) { return s == 0 ? zero_ggsw.data() : (s <= n_lwe ? bsk + (size_t)(s - 1) * ggsw_len : ksk); };
    auto mask_of = [&](unsigned s) { return s == 0 ? ct[n_lwe] : (s <= n_lwe ? ct[s - 1] : (u64)0); };
    // public inputs of the base proof: acc_init = (0, .., 0, testv) | counter 0 | accumulator 0 | hashes 0 | the cyclic circuit's verifier data
    std::vector<u64> base_pis(n_pi, 0);
    std::memcpy(base_pis.data() + kn - v->N, testv, 8 * (size_t)v->N);
    std::memcpy(base_pis.data() + n_pi - cyc.vk.size(), cyc.vk.data(), 8 * cyc.vk.size());

    struct Ready {   // owns the early phase's state until run_late consumes it (a failure on the way must not leak tens of MB of slot values)
        int buf = -1;
        vpbs_witness_state* state = nullptr;
        std::vector<u64> values, pis;
        Ready() = default;
        Ready(int b, std::vector<u64>&& v) : buf(b), values(std::move(v)) {}
        Ready(Ready&& o) noexcept : buf(o.buf), state(o.state), values(std::move(o.values)), pis(std::move(o.pis)) { o.state = nullptr; }
        Ready& operator=(Ready&& o) noexcept {
            if (this != &o) {
                drop();
                buf = o.buf; state = o.state; values = std::move(o.values); pis = std::move(o.pis);
                o.state = nullptr;
            }
            return *this;
        }
        Ready(const Ready&) = delete;
        Ready& operator=(const Ready&) = delete;
        void drop() {
            if (state) vpbs_witness_state_free(state);
            state = nullptr;
        }
        ~Ready() { drop(); }
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> free_bufs{0, 1, 2};
    std::deque<Ready> generated, ready;   // early thread -> uploader -> caller
    std::atomic<bool> failed{false};
    std::string thread_err;
    double t_early = 0;
    auto fail = [&](const std::string& m) {
        {   // the flag changes under the lock: a waiter that has just evaluated its predicate cannot miss the notification
            std::lock_guard<std::mutex> lk(mu);
            if (thread_err.empty()) thread_err = m;
            failed = true;
        }
        cv.notify_all();
    };
    std::thread early([&] {
        name_thread("vpbs-early");
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
            Ready r(b, std::vector<u64>(proof_words, 0));
            r.values.reserve(cyc.n_preset);
            r.values.insert(r.values.end(), pis_prev.begin(), pis_prev.end());
            r.values.push_back(s == 0 ? 0 : 1);   // condition: false only in the base step
            r.values.insert(r.values.end(), ggsw_of(s), ggsw_of(s) + ggsw_len);
            r.values.push_back(mask_of(s));
            r.values.insert(r.values.end(), cyc.vk.begin(), cyc.vk.end());
            r.values.insert(r.values.end(), dum.vk.begin(), dum.vk.end());
            r.values.insert(r.values.end(), v->dummy_proof.begin(), v->dummy_proof.end());
            r.values.insert(r.values.end(), n_pi, 0);   // the dummy proof's public inputs
            // a matrix this plan has filled before only gets its value-carrying positions rewritten
            const auto run_early = v->filled[b] ? vpbs_witness_plan_run_early_recycled : vpbs_witness_plan_run_early;
            if (run_early(cyc.plan, r.values.data(), 0, v->bufs[b], &r.state, e2, sizeof e2) != 0)
                return fail("early witness phase of step " + std::to_string(s) + ": " + e2);
            v->filled[b] = true;
            r.pis.resize(n_pi);
            for (size_t i = 0; i < n_pi; ++i) r.pis[i] = v->bufs[b][cyc.pi_pos[i]];   // public inputs never depend on the inner proof's words
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
        name_thread("vpbs-upload");
        for (unsigned s = 0; s < steps; ++s) {
            Ready r;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !generated.empty() || failed; });
                if (failed) return;
                r = std::move(generated.front());
                generated.pop_front();
            }
            if (vpbs_device_upload_bg(ctx, v->d_bufs[r.buf], v->bufs[r.buf], v->wire_words) != 0)
                return fail("upload of the early wires of step " + std::to_string(s));
            {
                std::lock_guard<std::mutex> lk(mu);
                ready.push_back(std::move(r));
            }
            cv.notify_all();
        }
    });
    auto stop = [&](const std::string& m, int rc) {
        fail(m);
        early.join();
        uploader.join();
        generated.clear();   // ~Ready frees the states nobody consumed
        ready.clear();
        std::string first;
        {
            std::lock_guard<std::mutex> lk(mu);
            first = thread_err;
        }
        say(first);
        return (long)rc;
    };

    // cyclic_base_proof (:292-299): a proof of the dummy circuit carrying the initial accumulator and the cyclic verifier data
    const double t0 = now();
    char e[256] = {0};
    vpbs_step_inputs in;
    vpbs_step_sizes sz{};
    if (vpbs_witness_plan_run(dum.plan, base_pis.data(), 0, v->base_wires, e, sizeof e) != 0) return stop(std::string("dummy witness: ") + e, VPBS_ERR_INVALID);
    dum.step_inputs(in, v->base_wires, false, base_pis.data());
    if (vpbs_step_sizes_get(ctx, &in, &sz) != 0 || 3 * sz.cap_words + sz.openings_words + sz.fri_words != proof_words)
        return stop("the proof of this shape does not have the number of words the cyclic circuit expects", VPBS_ERR_INVALID);
    std::vector<u64> proof(proof_words), pis;
    u64 *caps = proof.data(), *openings = caps + 3 * sz.cap_words, *fri = openings + sz.openings_words;   // the flat order of the proof targets
    int rc = dum.prove(in, caps, openings, fri);
    if (rc != 0) return stop(std::string("base proof: ") + vpbs_last_error(ctx), rc);
    const double t_base = now() - t0;
    if (v->step_fn) v->step_fn(v->step_user, 0);
    double t_late = 0, t_rows = 0, t_prove = 0;
    // the next step's late stages 1 and 2 run on a worker while this step's FRI stage is on the device (LateAhead): the prover reports its
    // sections (on_section, on this thread); the first report that finds the next step's early phase finished and uploaded takes it
    Ready next;   // declared before the worker: the worker is joined before the state it may be working on goes away
    bool have_next = false;
    LateAhead ahead;
    if (v->staged) {
        ahead.ranges = v->ahead_layout.ranges;
        ahead.packed = v->late_vals;
        ahead.start(cyc.plan);
    }
    struct Hook {
        LateAhead* ahead;
        std::mutex* mu;
        std::deque<Ready>* ready;
        Ready* next;
        bool* have_next;
        const u64* proof;
        static void section(void* user, int sec) {
            auto* h = static_cast<Hook*>(user);
            if (!*h->have_next) {
                std::lock_guard<std::mutex> lk(*h->mu);
                if (h->ready->empty()) return;   // its early phase is still running: this step's late phase runs whole, after the proof
                *h->next = std::move(h->ready->front());
                h->ready->pop_front();
                *h->have_next = true;
                h->ahead->begin(h->next->state, h->next->values.data(), h->proof);
            }
            h->ahead->post((unsigned)sec);
        }
    } hook{&ahead, &mu, &ready, &next, &have_next, proof.data()};
    // a rank of a SHARDED chain that cannot start a step's proof (its witness failed) takes part in that step's collectives on the failing
    // side: the other ranks' proofs of the step return VPBS_ERR_PEER and the chain ends on every rank instead of hanging on the others
    auto abandon_step = [&](int why) {
        if (!cyc.comm) return;
        vpbs_step_inputs shape;
        cyc.step_inputs(shape, nullptr, true, nullptr);
        (void)vpbs_prove_step_sharded_fail(ctx, &shape, cyc.comm, why);
    };
    for (unsigned s = 0; s < steps; ++s) {
        Ready r;
        if (have_next) {
            r = std::move(next);
            have_next = false;
        } else {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || failed; });
            if (failed) {
                lk.unlock();
                abandon_step(VPBS_ERR_INVALID);
                return stop("", VPBS_ERR_INVALID);
            }
            r = std::move(ready.front());
            ready.pop_front();
        }
        double t = now();
        if (ahead.active()) {   // stages that ran ahead on this step's state: wait for the last of them (usually long finished)
            std::string msg;
            if (!ahead.drain(msg)) {
                abandon_step(VPBS_ERR_INVALID);
                return stop("late witness phase of step " + std::to_string(s) + " (the previous proof does not verify in circuit): " + msg, VPBS_ERR_INVALID);
            }
        }
        std::copy(proof.begin(), proof.end(), r.values.begin());
        vpbs_witness_state* st = r.state;
        r.state = nullptr;
        rc = vpbs_witness_plan_run_late_packed(cyc.plan, st, r.values.data(), v->late_vals, e, sizeof e);   // consumes the state, also when it fails
        if (rc != 0) {
            abandon_step(rc);
            return stop("late witness phase of step " + std::to_string(s) + " (the previous proof does not verify in circuit): " + e, rc);
        }
        t_late += now() - t;
        t = now();
        rc = vpbs::device_scatter(ctx, v->d_bufs[r.buf], v->d_late_pos, v->late_vals, v->late_count, v->d_late_stage, false);   // queued ahead of the proof (see the device pipeline's loop)
        if (rc != 0) {
            abandon_step(rc);
            return stop(std::string("upload of the late wires: ") + vpbs_last_error(ctx), rc);
        }
        t_rows += now() - t;
        t = now();
        pis = std::move(r.pis);
        cyc.step_inputs(in, v->d_bufs[r.buf], true, pis.data());
        if (v->staged && s + 1 < steps) {
            in.on_section = &Hook::section;
            in.on_section_user = &hook;
        }
        rc = cyc.prove(in, caps, openings, fri);
        if (rc != 0) {
            if (ahead.active()) {   // the worker may be inside a stage of the next step's state: let it finish before that state is freed
                std::string ignored;
                (void)ahead.drain(ignored);
            }
            return stop("step " + std::to_string(s) + ": " + vpbs_last_error(ctx), rc);
        }
        t_prove += now() - t;
        {
            std::lock_guard<std::mutex> lk(mu);
            free_bufs.push_back(r.buf);
        }
        cv.notify_all();
        if (v->step_fn) v->step_fn(v->step_user, s + 1);
    }
    early.join();
    uploader.join();
    const double seconds = now() - t0;
    const long n_bytes = vpbs_step_proof_to_bytes(ctx, &in, cyc.n_const_cols, caps, openings, fri, proof_out, capacity);
    if (n_bytes <= 0) {
        say("the output buffer is too small for the proof");
        return VPBS_ERR_INVALID;
    }
    if (timing) {
        timing->seconds = seconds;
        timing->steps = steps;
        timing->base_proof_ms = 1e3 * t_base;
        timing->late_witness_ms = 1e3 * t_late / steps;
        timing->late_rows_upload_ms = 1e3 * t_rows / steps;
        timing->prove_step_ms = 1e3 * t_prove / steps;
        timing->early_witness_ms = 1e3 * t_early / steps;
        timing->late_ahead_ms = 1e3 * ahead.busy_s / steps;
    }
    return n_bytes;
}
}  // extern "C"
