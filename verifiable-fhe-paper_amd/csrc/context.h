// Prover context: one device, one HIP stream, a size-keyed device-memory pool (steady-state step proofs allocate
// nothing), cached twiddle / coset-power tables, and optional per-kernel HIP-event timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/vpbs_prover.h"
#include "kernels.h"

namespace vpbs {
struct DeviceError {
    int status;
    std::string what;
};

// The call that REPORTS a failure clears it: the waits only peek at the thread's pending error (vpbs::stream_sync), so that the
// stage-end checks still see a failed launch -- but once it has been turned into a DeviceError, and from there into the status and
// vpbs_last_error() of the entry point, it is consumed here.  A non-sticky failure (bad configuration, out of resources; this library's or
// another one's on the same thread) is thus reported by exactly one call; a C++ or Rust host needs no hipGetLastError of its own to go on.
#define VPBS_HIP(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            (void)hipGetLastError();                                                                           \
            throw ::vpbs::DeviceError{e_ == hipErrorOutOfMemory ? VPBS_ERR_OOM : VPBS_ERR_DEVICE,              \
                                      std::string(#expr) + ": " + hipGetErrorString(e_)};                      \
        }                                                                                                      \
    } while (0)
#define VPBS_REQUIRE(cond, msg)                                         \
    do {                                                                \
        if (!(cond)) throw ::vpbs::DeviceError{VPBS_ERR_INVALID, msg};  \
    } while (0)
}  // namespace vpbs

namespace vpbs {
// How a host thread waits for its stream.  hipStreamSynchronize spins: the thread that proves a chain burns a whole CPU while the device
// works (measured, VPBS_TRACE_IVC: with four chains on TWO CPUs a proving thread used 16 ms of CPU per proof inside vpbs_prove_step, of which
// about one is work).  In blocking mode the wait polls hipStreamQuery and sleeps in between.  Process-wide (it is about the CPUs the process
// has): vpbs_host_set_blocking_sync, default from VPBS_BLOCKING_SYNC, else AUTO -- block when the process may use fewer than 8 CPUs.
int blocking_sync_mode();                 // 0 spin, 1 block
hipError_t stream_sync(hipStream_t s);    // drop-in for hipStreamSynchronize
void stream_sync_forget(hipStream_t s);   // before hipStreamDestroy: the stream's completion word (api.hip) is freed
int sync_word_mode();                     // 1: waits read a word the device writes (default); 0: they go through the runtime
// out[k] = hash_no_pad(h_{k-1} || items[k]), h_{-1} = prefix; concurrent callers with long items share the lanes of the eight-lane host Poseidon
void hash_links_shared(const u64 prefix[4], const u64* const* items, size_t n_links, size_t item_len, u64* out);
void blocking_sync_budget_changed();      // the process's CPU budget was set: AUTO decides again
}  // namespace vpbs

struct vpbs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t upload_stream = nullptr;   // vpbs_device_upload_bg: host->device copies next to the work on `stream`
    unsigned log_n_max = 0, rate_bits = 3, cap_height = 4;
    vpbs::Tuning tune = vpbs::Tuning::from_env();   // vpbs_ctx_set_option
    vpbs_compat compat{0, 1, 1, 1};   // vpbs_compat_default: plonky2 0.2.0 as restated (include/vpbs_prover.h, the switch table)
    std::string err;

    // ---- device memory pool ----
    std::multimap<size_t, void*> free_blocks;
    std::unordered_map<void*, size_t> block_size;
    size_t pool_bytes = 0;
    void* alloc_bytes(size_t bytes);
    vpbs::u64* alloc_words(size_t words) { return static_cast<vpbs::u64*>(alloc_bytes(words * sizeof(vpbs::u64))); }
    void release(void* p);
    void trim();

    // ---- pinned staging for the small device->host results of the transcript (caps, openings, query records) ----
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // copy + stream synchronise; dst is ordinary (pageable) caller memory
    void d2h_sync(void* dst, const void* d_src, size_t bytes);
    void ensure_pinned();
    // a 4-byte device flag copied to pinned memory WITHOUT a synchronisation of its own: the value is there after the next d2h_sync (or
    // any synchronisation of the stream).  One flag in flight per context.  nullptr when no pinned memory could be had (the caller syncs).
    volatile unsigned* d2h_deferred_flag(const void* d_src);

    // ---- tables ----
    std::map<std::pair<unsigned, bool>, vpbs::u64*> root_tables;                 // (log_n, inverse)
    std::map<std::tuple<unsigned, unsigned, vpbs::u64>, vpbs::u64*> prescale_tables;  // (log_n, rate_bits, shift)
    std::map<std::tuple<unsigned, unsigned, vpbs::u64>, vpbs::u64*> lde_tables;       // (log_n, rate_bits, shift)
    std::map<unsigned, vpbs::u64*> ring_tables;  // log_N -> [ROOTS | INVROOTS] of the negacyclic NTT (params_{N}.rs), 2N words
    const vpbs::u64* ring_table(unsigned log_n_ring);
    std::map<unsigned, vpbs::u64*> l0_tables;  // log_n -> L_0 on the coset, leaf order
    const vpbs::u64* l0_table(unsigned log_n);
    const vpbs::u64* roots(unsigned log_n, bool inverse);
    const vpbs::u64* prescale(unsigned log_n, unsigned rate_bits, vpbs::u64 shift);    // plain powers (shift w_big^r)^i
    const vpbs::u64* lde_table(unsigned log_n, unsigned rate_bits, vpbs::u64 shift);   // what launch_coset_lde reads (kernels.h)

    // ---- helper streams for the gate-constraint kernels (created on first use) ----
    hipStream_t gate_streams[2] = {nullptr, nullptr};
    hipEvent_t gate_fork = nullptr, gate_join[2] = {nullptr, nullptr};
    void ensure_gate_lanes();
    // 1 (one stream): the default since the gate constraints run in the LDS-tile kernel -- its workgroups fill a CU's LDS and registers, so
    // the permutation part gains nothing beside it (9.67 vs 9.93 ms per step proof); 3: the helper streams of rounds 1-2, still the better
    // arrangement for the per-gate launches.  VPBS_GATE_LANES=3 in the environment makes that the default; vpbs_ctx_set_gate_lanes.
    unsigned gate_lanes = default_gate_lanes();
    static unsigned default_gate_lanes() {
        const char* e = getenv("VPBS_GATE_LANES");
        return e && atoi(e) == 3 ? 3u : 1u;
    }

    // ---- timing ----
    vpbs::u64* d_clock_samples = nullptr;   // [CLOCK_SAMPLES][2], one pair per leaf-hash launch while timing is on (ring)
    static constexpr unsigned CLOCK_SAMPLES = 1024;
    unsigned clock_samples = 0;
    vpbs::u64* next_clock_sample();         // nullptr unless timing is on
    bool timing = false;
    std::string timing_only;  // when non-empty, only this timer is recorded
    struct Pending {
        int name_id;
        hipEvent_t start, stop;
    };
    std::vector<std::string> timer_names;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, std::pair<double, long>> totals;
    int timer_id(const char* name);
    hipEvent_t get_event();
    void resolve_timing();
};

namespace vpbs {
// RAII: HIP events around a group of launches on the ctx stream (no-op unless timing is enabled)
struct Timed {
    vpbs_ctx* c;
    int id = -1;
    hipEvent_t start = nullptr;
    Timed(vpbs_ctx* ctx, const char* name) : c(ctx) {
        if (!c->timing) return;
        if (!c->timing_only.empty() && c->timing_only != name) return;
        id = c->timer_id(name);
        start = c->get_event();
        (void)hipEventRecord(start, c->stream);
    }
    ~Timed() {
        if (id < 0) return;
        hipEvent_t stop = c->get_event();
        (void)hipEventRecord(stop, c->stream);
        c->pending.push_back({id, start, stop});
    }
};
}  // namespace vpbs

struct vpbs_batch {
    vpbs_ctx* ctx = nullptr;
    unsigned ncols = 0, log_n = 0;
    vpbs::u64* d_coeffs = nullptr;   // [ncols][n]
    vpbs::u64* d_lde = nullptr;      // [ncols][n << rate_bits], leaf order
    vpbs::u64* d_digests = nullptr;  // Merkle levels back to back
    std::vector<size_t> level_off;   // word offset of each level; last level = cap
    // coset sharding: this batch holds leaf blocks [shard * per, (shard + 1) * per) of the 2^rate_bits
    // blocks, per = 2^rate_bits / n_shards, i.e. the leaves [leaf_offset(), leaf_offset() + lde_len()) and the
    // cap entries [shard * cap_len(), (shard + 1) * cap_len()).  n_shards == 1: the whole commitment.
    unsigned shard = 0, n_shards = 1;
    size_t n() const { return (size_t)1 << log_n; }
    unsigned blocks() const { return (1u << ctx->rate_bits) / n_shards; }
    size_t lde_len() const { return n() * blocks(); }
    size_t leaf_offset() const { return (size_t)shard * lde_len(); }
    size_t cap_len() const { return ((size_t)1 << ctx->cap_height) / n_shards; }
    unsigned n_levels() const { return (unsigned)level_off.size(); }
};

namespace vpbs {
// Merkle level layout for a tree with n_leaves leaves and cap height h: returns total words
size_t merkle_layout(size_t n_leaves, unsigned cap_height, std::vector<size_t>& level_off);
// commit a device-resident matrix (values or coefficients); returns a new batch
vpbs_batch* commit_device(vpbs_ctx* ctx, const u64* d_in, unsigned ncols, unsigned log_n, bool is_values, unsigned shard = 0,
                          unsigned n_shards = 1);
void batch_cap_to_host(vpbs_batch* b, u64* cap_out);
// vpbs_device_scatter's copy + kernel on the context's stream; wait = false: queued only (api.hip)
int device_scatter(vpbs_ctx* c, uint64_t* d_dst, const uint64_t* d_positions, const uint64_t* host_values, size_t count, uint64_t* d_stage, bool wait);
}  // namespace vpbs
