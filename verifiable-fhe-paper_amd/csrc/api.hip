// extern "C" surface: context, PolynomialBatch handles, kernel-level hooks.  See include/vpbs_prover.h for the
// plonky2 function each entry point replaces.
#include <atomic>
#include <sys/prctl.h>
#include <time.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>

#include "context.h"
#include "poseidon.h"
#include "host/poseidon_x8.h"

using vpbs::DeviceError;
using vpbs::u64;

// ---------------- ctx internals ----------------
void* vpbs_ctx::alloc_bytes(size_t bytes) {
    if (bytes == 0) bytes = 8;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = free_blocks.find(bytes);
    if (it != free_blocks.end()) {
        void* p = it->second;
        free_blocks.erase(it);
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        trim();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) throw DeviceError{VPBS_ERR_OOM, "hipMalloc(" + std::to_string(bytes) + ") failed"};
    block_size[p] = bytes;
    pool_bytes += bytes;
    return p;
}
void vpbs_ctx::release(void* p) {
    if (!p) return;
    auto it = block_size.find(p);
    if (it != block_size.end()) free_blocks.emplace(it->second, p);  // unknown pointers are ignored, never thrown on
}
void vpbs_ctx::trim() {
    (void)vpbs::stream_sync(stream);
    for (auto& kv : free_blocks) {
        pool_bytes -= kv.first;
        block_size.erase(kv.second);
        (void)hipFree(kv.second);
    }
    free_blocks.clear();
}
namespace vpbs {
std::atomic<int> g_blocking_sync{-1};   // vpbs_host_set_blocking_sync: -1 = not set (environment, then AUTO)
std::atomic<int> g_blocking_auto{-1};   // AUTO's answer, computed once per CPU budget (vpbs_host_set_cpu_budget resets it)
int blocking_sync_mode() {
    const int m = g_blocking_sync.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    static const int from_env = [] {
        const char* e = getenv("VPBS_BLOCKING_SYNC");
        return e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    if (from_env >= 0) return from_env;
    int a = g_blocking_auto.load(std::memory_order_relaxed);
    if (a < 0) {
        a = vpbs_host_cpu_budget() < 8 ? 1 : 0;   // AUTO: a process with few CPUs cannot afford a spinning thread per chain
        g_blocking_auto.store(a, std::memory_order_relaxed);
    }
    return a;
}
std::atomic<int> g_budget_cached{-1};    // the CPU budget as AUTO last saw it (read per wait)
void blocking_sync_budget_changed() {
    g_blocking_auto.store(-1, std::memory_order_relaxed);
    g_budget_cached.store(-1, std::memory_order_relaxed);
}
// One wait's answer.  Explicit setting and environment as blocking_sync_mode(); AUTO sleeps below 8 CPUs, and above that whenever the threads
// waiting for the device right now are more than a quarter of the CPUs: spinning is for the ONE chain that wants its latency, not for eight
// proving threads on ten CPUs (measured: 0.137-0.141 vPBS/s spinning, 0.153-0.157 sleeping, 80 against 20-27 CPU-ms per proof;
// tools/experiments/blocking_at_8_cpus.sh).  Six chains on 16 CPUs sleep as well: same throughput, half the CPU time.
std::atomic<int> g_sync_waiters{0};
static bool this_wait_sleeps(int waiters) {
    if (g_blocking_sync.load(std::memory_order_relaxed) >= 0 || blocking_sync_mode()) return blocking_sync_mode() != 0;
    static const bool env_says_spin = getenv("VPBS_BLOCKING_SYNC") != nullptr;   // set to 0: blocking_sync_mode() returned 0 because of it
    if (env_says_spin) return false;
    int budget = g_budget_cached.load(std::memory_order_relaxed);
    if (budget < 0) {
        budget = (int)vpbs_host_cpu_budget();
        g_budget_cached.store(budget, std::memory_order_relaxed);
    }
    return waiters * 4 > budget;
}

// ---- completion words ----
// Every wait INSIDE the runtime (hipStreamSynchronize, and equally the first hipStreamQuery on a busy stream) hands the stream's last command
// to the HSA async-events thread, and that thread then busy-waits through KFD event ioctls until the command retires: one whole CPU for as
// long as the device has work, next to the waiting thread itself (tools/experiments/graph_cpu_probe.hip: 8.5 ms of CPU per 8.9 ms of device
// work, graph launch or not; 0.5 ms when nobody asks the runtime).  So the library does not ask: a one-thread kernel behind the work writes a
// sequence number into a word of host memory the device has mapped, and the host reads that word -- spinning, or napping in blocking mode.
// The copies ahead of it on the stream have landed when the word changes (stream order; the kernel fences at system scope before it writes).
struct SyncWord {
    volatile u64* host = nullptr;   // hipHostMalloc, mapped: the device writes, the host reads
    u64* dev = nullptr;
    std::mutex mu;                  // a sequence number and the launch of its marker are one step
    u64 seq = 0;
};
__global__ void sync_word_kernel(volatile u64* word, u64 seq) {
    __threadfence_system();
    *word = seq;
}
std::mutex g_sync_words_mu;
std::unordered_map<hipStream_t, std::unique_ptr<SyncWord>> g_sync_words;
std::atomic<int> g_sync_word_mode{-1};   // vpbs_host_set_sync_word: -1 = environment (VPBS_SYNC_WORD), default on
// VPBS_TRACE_SYNC: at exit, the waits that went through completion words and how many of them the 200 ms look at the runtime had to end
// (a stream that was through without its word having moved: never seen; a soak prints 0)
std::atomic<unsigned long> g_sync_waits{0}, g_sync_rescued{0};
const bool g_sync_trace = [] {
    if (!getenv("VPBS_TRACE_SYNC")) return false;
    atexit([] { fprintf(stderr, "[sync] %lu waits on completion words, %lu ended by the runtime check\n", g_sync_waits.load(), g_sync_rescued.load()); });
    return true;
}();
int sync_word_mode() {
    const int m = g_sync_word_mode.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    static const int from_env = [] {
        const char* e = getenv("VPBS_SYNC_WORD");
        return e ? (atoi(e) != 0 ? 1 : 0) : 1;
    }();
    return from_env;
}
static SyncWord* sync_word_of(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_sync_words_mu);
    auto& slot = g_sync_words[s];
    if (!slot) {
        std::unique_ptr<SyncWord> w(new SyncWord());
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return nullptr;
        *static_cast<volatile u64*>(h) = 0;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipHostFree(h);
            return nullptr;
        }
        w->host = static_cast<volatile u64*>(h);
        w->dev = static_cast<u64*>(d);
        slot = std::move(w);
    }
    return slot.get();
}
void stream_sync_forget(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_sync_words_mu);
    auto it = g_sync_words.find(s);
    if (it == g_sync_words.end()) return;
    if (it->second && it->second->host) (void)hipHostFree(const_cast<u64*>(it->second->host));
    g_sync_words.erase(it);
}
// a nap of a sleeping wait: 30 us at first, then a sixteenth of the time already waited, at most 200 us -- a wait ends at most ~6 % late, and a
// wait of several milliseconds (six chains sharing the device) costs some forty wake-ups instead of a hundred
static void nap(long waited_ns) {
    static thread_local bool slack_set = false;
    if (!slack_set) {
        (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);   // 1 us instead of the default 50 us: a nap is a nap
        slack_set = true;
    }
    timespec ts{0, std::min(200000L, std::max(30000L, waited_ns / 16))};
    (void)nanosleep(&ts, nullptr);
}
static bool launch_class_error(hipError_t e) {
    switch (e) {
        case hipErrorInvalidConfiguration:
        case hipErrorLaunchOutOfResources:
        case hipErrorLaunchFailure:
        case hipErrorInvalidDeviceFunction:
        case hipErrorIllegalAddress:
        case hipErrorNoBinaryForGpu:
        case hipErrorSharedObjectInitFailed:
        case hipErrorLaunchTimeOut:
            return true;
        default:
            return false;
    }
}
hipError_t stream_sync(hipStream_t s) {
    struct Waiter {
        int n;
        Waiter() : n(g_sync_waiters.fetch_add(1, std::memory_order_relaxed) + 1) {}
        ~Waiter() { g_sync_waiters.fetch_sub(1, std::memory_order_relaxed); }
    } waiter;
    const bool block = this_wait_sleeps(waiter.n);
    if (sync_word_mode()) {
        if (SyncWord* w = sync_word_of(s)) {
            // A launch that failed earlier on this thread (bad configuration, out of resources) left its error pending and the buffers it should
            // have written untouched: the wait reports it instead of waiting for work that was never queued -- and it must not CONSUME the
            // thread's error state either (hipGetLastError here once hid such failures from the checks at the end of the stages: ADVICE r04).
            // Only errors a launch or the device raises count: other calls of the process (a peer-access probe of the collective library, a
            // stream query that answered "not ready") may leave benign codes behind which are not this library's to judge.
            if (const hipError_t pending = hipPeekAtLastError(); launch_class_error(pending)) return pending;
            // the sequence number is taken and its marker queued as one step: two threads waiting on the same stream see their markers in
            // the order of their numbers, so the word never moves backwards and nobody returns before its own marker
            u64 seq;
            hipError_t launched;
            {
                std::lock_guard<std::mutex> lk(w->mu);
                seq = ++w->seq;
                u64* word = w->dev;
                void* args[] = {&word, &seq};
                launched = hipLaunchKernel(reinterpret_cast<const void*>(&sync_word_kernel), dim3(1), dim3(1), args, 0, s);
            }
            if (launched == hipSuccess) {
                // the runtime is asked only once in a long while: a stream that faulted never writes its word
                const auto t_begin = std::chrono::steady_clock::now();
                auto t_check = t_begin + std::chrono::milliseconds(200);
                long waited_ns = 0;
                g_sync_waits.fetch_add(1, std::memory_order_relaxed);
                for (unsigned i = 0;; ++i) {
                    if (__atomic_load_n(w->host, __ATOMIC_ACQUIRE) >= seq) return hipSuccess;   // acquire: the copies in front of the marker are visible
                    if (block && i >= 64) nap(waited_ns);
                    else __builtin_ia32_pause();
                    if ((i & 0xff) == 0xff || (block && i >= 64)) {
                        const auto now = std::chrono::steady_clock::now();
                        waited_ns = (long)std::chrono::duration_cast<std::chrono::nanoseconds>(now - t_begin).count();
                        if (now >= t_check) {
                            const hipError_t q = hipStreamQuery(s);
                            if (q != hipErrorNotReady && __atomic_load_n(w->host, __ATOMIC_ACQUIRE) < seq) {   // the stream is through (or broken) and the word has not moved
                                g_sync_rescued.fetch_add(1, std::memory_order_relaxed);
                                return q == hipSuccess ? hipStreamSynchronize(s) : q;
                            }
                            t_check = now + std::chrono::milliseconds(200);
                        }
                    }
                }
            }
        }
    }
    if (!block) return hipStreamSynchronize(s);
    // Sleeping wait through the runtime (VPBS_SYNC_WORD=0): poll the stream and give the CPU away in between.  (An event made with
    // hipEventBlockingSync does not do it on this runtime: measured, the waiting thread still used 100 % of a CPU.)  A short spin first, then
    // naps of 30 us: a wait ends at most ~50 us late, eleven times per step proof, which a GPU shared by several chains does not notice.
    for (unsigned i = 0;; ++i) {
        const hipError_t q = hipStreamQuery(s);
        if (q != hipErrorNotReady) return q;
        if (i < 64) continue;
        nap(0);
    }
}
}  // namespace vpbs

// ---- links of hash chains for several callers at once ----
// The bootstrapping-key hash chain of a vPBS is 730 x 2 049 DEPENDENT permutations (h_s = hash_no_pad(h_{s-1} || GGSW_s),
// ivc_based_vpbs.rs:64-78): one scalar permutation after the other, 2.3 ms of CPU per chained proof -- a sixth of what a rank with two CPUs
// has.  The chains of ONE process are independent of each other, though: callers that ask for the same number of links while a batch is
// being hashed are taken together by the next batch, one chain per AVX-512 lane (0.41 instead of 1.28 us per permutation with eight of
// them), all their links in lockstep.  Whoever finds no batch in progress runs the next one for everybody waiting (no thread of its own);
// fewer than three jobs go one by one.  For processes that are short of CPUs only (see hash_links_shared).
namespace vpbs {
namespace {
struct SpongeJob {
    const u64* prefix;          // [4]
    const u64* const* items;    // [n_links] pointers to item_len words each
    size_t n_links, len;
    u64* out;                   // [n_links][4]
    bool done = false;
};
void sponge_scalar(const SpongeJob& j) {
    const u64* h = j.prefix;
    for (size_t k = 0; k < j.n_links; ++k) {
        u64 s[12] = {0};
        const size_t total = 4 + j.len;
        for (size_t off = 0; off < total; off += 8) {
            const size_t blk = total - off < 8 ? total - off : 8;
            for (size_t i = 0; i < blk; ++i) s[i] = off + i < 4 ? h[off + i] : j.items[k][off + i - 4];
            poseidon::permute_host(s);
        }
        for (int i = 0; i < 4; ++i) j.out[4 * k + i] = s[i];
        h = j.out + 4 * k;
    }
}
#if defined(VPBS_HAVE_POSEIDON_X8)
std::mutex g_sponge_mu;
std::condition_variable g_sponge_cv;
std::vector<SpongeJob*> g_sponge_queue;
bool g_sponge_busy = false;
size_t g_sponge_last = 1, g_sponge_prev = 0;   // jobs in the last batch and in the one before
__attribute__((target("avx512f,avx512dq"))) void sponge_lanes(SpongeJob* const* jobs, unsigned cnt) {   // 3 <= cnt <= 8 jobs of one shape
    using poseidon_x8::V;
    const size_t total = 4 + jobs[0]->len, n_links = jobs[0]->n_links;
    alignas(64) u64 buf[8][8];
    const SpongeJob* lane[8];
    for (unsigned l = 0; l < 8; ++l) lane[l] = jobs[l < cnt ? l : 0];   // spare lanes repeat the first job
    for (size_t k = 0; k < n_links; ++k) {
        const u64 *item[8], *prefix[8];
        for (unsigned l = 0; l < 8; ++l) {
            item[l] = lane[l]->items[k];
            prefix[l] = k ? lane[l]->out + 4 * (k - 1) : lane[l]->prefix;
        }
        V s[12];
        for (int i = 0; i < 12; ++i) s[i] = _mm512_setzero_si512();
        for (size_t off = 0; off < total; off += 8) {
            const size_t blk = total - off < 8 ? total - off : 8;
            if (off == 0 || blk < 8) {   // the block with the prefix, a short last block: word by word
                for (unsigned l = 0; l < 8; ++l)
                    for (size_t i = 0; i < blk; ++i) buf[i][l] = off + i < 4 ? prefix[l][off + i] : item[l][off + i - 4];
                for (size_t i = 0; i < blk; ++i) s[i] = _mm512_load_si512(buf[i]);   // overwrite mode: the rate words are replaced
            } else {   // eight words of every lane's item in one load each, transposed in registers (lane-major -> word-major)
                V r[8], t[8], v[8];
                for (unsigned l = 0; l < 8; ++l) {
                    r[l] = _mm512_loadu_si512(item[l] + off - 4);
                    _mm_prefetch(reinterpret_cast<const char*>(item[l] + off - 4 + 64), _MM_HINT_T0);   // eight blocks ahead
                }
                for (int q = 0; q < 4; ++q) {
                    t[2 * q] = _mm512_unpacklo_epi64(r[2 * q], r[2 * q + 1]);
                    t[2 * q + 1] = _mm512_unpackhi_epi64(r[2 * q], r[2 * q + 1]);
                }
                for (int odd = 0; odd < 2; ++odd) {
                    v[4 * odd + 0] = _mm512_shuffle_i64x2(t[0 + odd], t[2 + odd], 0x88);
                    v[4 * odd + 1] = _mm512_shuffle_i64x2(t[0 + odd], t[2 + odd], 0xdd);
                    v[4 * odd + 2] = _mm512_shuffle_i64x2(t[4 + odd], t[6 + odd], 0x88);
                    v[4 * odd + 3] = _mm512_shuffle_i64x2(t[4 + odd], t[6 + odd], 0xdd);
                    s[0 + odd] = _mm512_shuffle_i64x2(v[4 * odd + 0], v[4 * odd + 2], 0x88);
                    s[4 + odd] = _mm512_shuffle_i64x2(v[4 * odd + 0], v[4 * odd + 2], 0xdd);
                    s[2 + odd] = _mm512_shuffle_i64x2(v[4 * odd + 1], v[4 * odd + 3], 0x88);
                    s[6 + odd] = _mm512_shuffle_i64x2(v[4 * odd + 1], v[4 * odd + 3], 0xdd);
                }
            }
            poseidon_x8::permute(s, nullptr);
        }
        for (int i = 0; i < 4; ++i) {
            _mm512_store_si512(buf[i], s[i]);
            for (unsigned l = 0; l < cnt; ++l) jobs[l]->out[4 * k + i] = buf[i][l];
        }
    }
}
#endif
}  // namespace

// out[k] = hash_no_pad(h_{k-1} || items[k]), h_{-1} = prefix, k < n_links
void hash_links_shared(const u64 prefix[4], const u64* const* items, size_t n_links, size_t item_len, u64* out) {
    if (n_links == 0) return;
    SpongeJob job{prefix, items, n_links, item_len, out};
#if defined(VPBS_HAVE_POSEIDON_X8)
    // only where CPU time is what the process is short of (the switch of the sleeping waits: AUTO below 8 CPUs): a batch runs on ONE thread
    // while the other callers sleep, so with CPUs to spare every caller hashing its own links at once is sooner done
    if (poseidon_x8::enabled() && item_len >= 64 && blocking_sync_mode()) {
        std::unique_lock<std::mutex> lk(g_sponge_mu);
        g_sponge_queue.push_back(&job);
        g_sponge_cv.notify_all();   // a caller collecting its batch may be waiting for this one
        while (!job.done) {
            if (g_sponge_busy) {
                g_sponge_cv.wait(lk);
                continue;
            }
            g_sponge_busy = true;   // this caller runs the next batch: up to eight waiting jobs of the first one's shape
            // the callers of the last batch were all released at once and are on their way back: the first one here gives the others a
            // moment (a hundredth of what the batch will take, 50 us .. 1 ms) before it settles for fewer lanes.  The last TWO batches
            // together say how many to expect: otherwise two groups of four keep alternating, each hashing while the other waits
            const size_t expect = std::min<size_t>(8, g_sponge_last + g_sponge_prev);
            if (g_sponge_queue.size() < expect) {
                const long us = std::min<long>(1000, std::max<long>(50, (long)(n_links * (item_len / 8) * 3 / 100)));
                g_sponge_cv.wait_for(lk, std::chrono::microseconds(us), [expect] { return g_sponge_queue.size() >= expect; });
            }
            SpongeJob* batch[8];
            unsigned cnt = 0;
            const size_t len = g_sponge_queue.front()->len, links = g_sponge_queue.front()->n_links;
            for (auto it = g_sponge_queue.begin(); it != g_sponge_queue.end() && cnt < 8;) {
                if ((*it)->len == len && (*it)->n_links == links) {
                    batch[cnt++] = *it;
                    it = g_sponge_queue.erase(it);
                } else {
                    ++it;
                }
            }
            lk.unlock();
            if (cnt >= 3) sponge_lanes(batch, cnt);
            else
                for (unsigned i = 0; i < cnt; ++i) sponge_scalar(*batch[i]);
            lk.lock();
            for (unsigned i = 0; i < cnt; ++i) batch[i]->done = true;
            g_sponge_prev = g_sponge_last;
            g_sponge_last = cnt;
            g_sponge_busy = false;
            g_sponge_cv.notify_all();
        }
        return;
    }
#endif
    sponge_scalar(job);
}
}  // namespace vpbs

extern "C" int vpbs_host_set_blocking_sync(int on) {
    vpbs::g_blocking_sync.store(on < 0 ? -1 : (on ? 1 : 0));
    return vpbs::blocking_sync_mode();
}
extern "C" int vpbs_host_blocking_sync(void) { return vpbs::blocking_sync_mode(); }
extern "C" int vpbs_host_set_sync_word(int on) {
    vpbs::g_sync_word_mode.store(on < 0 ? -1 : (on ? 1 : 0));
    return vpbs::sync_word_mode();
}

static double trace_now_us() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
void vpbs_ctx::ensure_pinned() {
    constexpr size_t STAGE = (size_t)1 << 20;
    if (pinned) return;
    if (hipHostMalloc(&pinned, STAGE, hipHostMallocDefault) == hipSuccess) pinned_bytes = STAGE - 64;   // the last 64 bytes: deferred flags
    else pinned = nullptr;
}
volatile unsigned* vpbs_ctx::d2h_deferred_flag(const void* d_src) {
    ensure_pinned();
    if (!pinned) return nullptr;
    auto* slot = reinterpret_cast<volatile unsigned*>(static_cast<char*>(pinned) + pinned_bytes);
    *slot = 0xFFFFFFFFu;
    VPBS_HIP(hipMemcpyAsync(const_cast<unsigned*>(slot), d_src, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    return slot;
}
void vpbs_ctx::d2h_sync(void* dst, const void* d_src, size_t bytes) {
    static const bool trace = getenv("VPBS_TRACE") != nullptr;
    const double t_begin = trace ? trace_now_us() : 0;
    struct Tr {
        bool on; double t0; size_t b;
        ~Tr() { if (on) std::fprintf(stderr, "[vpbs trace] d2h_sync %zu B: host arrived %.1f us, returned %.1f us (waited %.1f)\n", b, t0, trace_now_us(), trace_now_us() - t0); }
    } tr{trace, t_begin, bytes};
    ensure_pinned();
    if (pinned && bytes <= pinned_bytes) {
        VPBS_HIP(hipMemcpyAsync(pinned, d_src, bytes, hipMemcpyDeviceToHost, stream));
        VPBS_HIP(vpbs::stream_sync(stream));
        std::memcpy(dst, pinned, bytes);
    } else {
        VPBS_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream));
        VPBS_HIP(vpbs::stream_sync(stream));
    }
}
void vpbs_ctx::ensure_gate_lanes() {
    if (gate_fork) return;
    for (auto& st : gate_streams) VPBS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    VPBS_HIP(hipEventCreateWithFlags(&gate_fork, hipEventDisableTiming));
    for (auto& e : gate_join) VPBS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
}
const u64* vpbs_ctx::roots(unsigned log_n, bool inverse) {
    auto key = std::make_pair(log_n, inverse);
    auto it = root_tables.find(key);
    if (it != root_tables.end()) return it->second;
    u64* t = alloc_words(vpbs::root_table_words(log_n));  // all n powers + the radix-16 rounds' block twiddle tables (ntt.hip root_table_words)
    vpbs::launch_root_table(stream, t, log_n, inverse);
    root_tables[key] = t;
    return t;
}
const u64* vpbs_ctx::ring_table(unsigned log_n_ring) {
    auto it = ring_tables.find(log_n_ring);
    if (it != ring_tables.end()) return it->second;
    const size_t n = (size_t)1 << log_n_ring;
    std::vector<u64> h(2 * n);
    u64 ninv;
    if (vpbs_ntt_params(log_n_ring, h.data(), h.data() + n, &ninv) != 0) throw DeviceError{VPBS_ERR_INVALID, "unsupported ring dimension"};
    u64* t = alloc_words(2 * n);
    VPBS_HIP(hipMemcpyAsync(t, h.data(), sizeof(u64) * 2 * n, hipMemcpyHostToDevice, stream));
    VPBS_HIP(vpbs::stream_sync(stream));
    ring_tables[log_n_ring] = t;
    return t;
}
const u64* vpbs_ctx::l0_table(unsigned log_n) {
    auto it = l0_tables.find(log_n);
    if (it != l0_tables.end()) return it->second;
    u64* t = alloc_words((size_t)1 << (log_n + rate_bits));
    vpbs::launch_l0_table(stream, roots(log_n + rate_bits, false), log_n, rate_bits, t);
    l0_tables[log_n] = t;
    return t;
}
const u64* vpbs_ctx::prescale(unsigned log_n, unsigned rb, u64 shift) {
    auto key = std::make_tuple(log_n, rb, shift);
    auto it = prescale_tables.find(key);
    if (it != prescale_tables.end()) return it->second;
    u64* t = alloc_words((size_t)1 << (log_n + rb));
    vpbs::launch_prescale_table(stream, t, log_n, rb, shift);
    prescale_tables[key] = t;
    return t;
}
const u64* vpbs_ctx::lde_table(unsigned log_n, unsigned rb, u64 shift) {
    auto key = std::make_tuple(log_n, rb, shift);
    auto it = lde_tables.find(key);
    if (it != lde_tables.end()) return it->second;
    u64* t = alloc_words(vpbs::lde_table_words(log_n, rb));
    vpbs::launch_lde_table(stream, t, log_n, rb, shift);
    lde_tables[key] = t;
    return t;
}
int vpbs_ctx::timer_id(const char* name) {
    for (size_t i = 0; i < timer_names.size(); ++i)
        if (timer_names[i] == name) return (int)i;
    timer_names.emplace_back(name);
    return (int)timer_names.size() - 1;
}
hipEvent_t vpbs_ctx::get_event() {
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void vpbs_ctx::resolve_timing() {
    if (pending.empty()) return;
    (void)vpbs::stream_sync(stream);
    for (auto& p : pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            auto& t = totals[timer_names[p.name_id]];
            t.first += ms;
            t.second += 1;
        }
        event_pool.push_back(p.start);
        event_pool.push_back(p.stop);
    }
    pending.clear();
}

namespace vpbs {
size_t merkle_layout(size_t n_leaves, unsigned cap_height, std::vector<size_t>& level_off) {
    unsigned log_leaves = 0;
    while (((size_t)1 << log_leaves) < n_leaves) ++log_leaves;
    VPBS_REQUIRE(((size_t)1 << log_leaves) == n_leaves, "leaf count must be a power of two");
    VPBS_REQUIRE(log_leaves >= cap_height, "tree smaller than its cap");
    level_off.clear();
    size_t off = 0;
    for (unsigned k = 0; k + cap_height <= log_leaves; ++k) {
        level_off.push_back(off);
        off += 4 * (n_leaves >> k);
    }
    return off;
}

vpbs_batch* commit_device(vpbs_ctx* ctx, const u64* d_in, unsigned ncols, unsigned log_n, bool is_values, unsigned shard,
                          unsigned n_shards) {
    VPBS_REQUIRE(ncols > 0, "ncols == 0");
    VPBS_REQUIRE(log_n <= ctx->log_n_max, "log_n exceeds the context's log_n_max");
    VPBS_REQUIRE(n_shards >= 1 && (n_shards & (n_shards - 1)) == 0 && n_shards <= (1u << ctx->rate_bits) && shard < n_shards,
                 "n_shards must be a power of two <= 2^rate_bits and shard < n_shards");
    VPBS_REQUIRE(((size_t)1 << ctx->cap_height) >= n_shards, "cap smaller than the shard count");
    auto* b = new vpbs_batch();
    b->ctx = ctx;
    b->ncols = ncols;
    b->log_n = log_n;
    b->shard = shard;
    b->n_shards = n_shards;
    const size_t n = b->n(), L = b->lde_len();
    try {
        b->d_coeffs = ctx->alloc_words((size_t)ncols * n);
        b->d_lde = ctx->alloc_words((size_t)ncols * L);
        // a shard owns whole cap subtrees: its local tree stops at cap_len() roots, i.e. local cap height
        unsigned local_cap_h = ctx->cap_height;
        for (unsigned k = n_shards; k > 1; k >>= 1) --local_cap_h;
        const size_t dig_words = merkle_layout(L, local_cap_h, b->level_off);
        b->d_digests = ctx->alloc_words(dig_words);
        if (is_values) {
            Timed t(ctx, "intt");
            launch_intt(ctx->stream, d_in, b->d_coeffs, b->d_lde /* scratch */, ctx->roots(log_n, true), ncols, log_n);
        } else {
            VPBS_HIP(hipMemcpyAsync(b->d_coeffs, d_in, sizeof(u64) * ncols * n, hipMemcpyDeviceToDevice, ctx->stream));
        }
        const u64* roots = ctx->roots(log_n, false);
        const u64* ps = ctx->lde_table(log_n, ctx->rate_bits, gl::GENERATOR);
        {
            Timed t(ctx, "coset_lde");
            launch_coset_lde(ctx->stream, b->d_coeffs, b->d_lde, roots, ps, ncols, log_n, ctx->rate_bits, shard * b->blocks(),
                             b->blocks());
        }
        // (Measured and rejected, round 2: hashing the leaves in two or four contiguous ranges and letting the subtrees of a finished range
        // climb on a helper stream while the next range is hashed.  The leaf kernel's waves occupy every wave slot for a whole generation
        // (~1.9 ms with 135 columns), so the helper stream's kernels only start when the next range drains: 10.36 / 11.46 ms per step proof
        // with 2 / 4 ranges against 10.01 ms with one launch.)
        {
            Timed t(ctx, "leaf_hash");
            launch_leaf_hash(ctx->stream, b->d_lde, ncols, L, L, b->d_digests, ctx->next_clock_sample());
        }
        {
            Timed t(ctx, "merkle_levels");
            launch_merkle_tree(ctx->stream, ctx->tune, b->d_digests, b->level_off.data(), b->n_levels(), L);
        }
        VPBS_HIP(hipGetLastError());
    } catch (...) {
        vpbs_batch_free(b);
        throw;
    }
    return b;
}

void batch_cap_to_host(vpbs_batch* b, u64* cap_out) {
    b->ctx->d2h_sync(cap_out, b->d_digests + b->level_off.back(), sizeof(u64) * 4 * b->cap_len());
}
}  // namespace vpbs

// ---------------- error plumbing ----------------
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

namespace {
// temporary device copy of a host matrix
struct DevTemp {
    vpbs_ctx* c;
    u64* p;
    DevTemp(vpbs_ctx* ctx, const u64* host, size_t words) : c(ctx), p(ctx->alloc_words(words)) {
        if (host) {
            hipError_t e = hipMemcpyAsync(p, host, words * sizeof(u64), hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) {
                c->release(p);
                throw DeviceError{VPBS_ERR_DEVICE, std::string("H2D copy: ") + hipGetErrorString(e)};
            }
        }
    }
    ~DevTemp() {
        (void)vpbs::stream_sync(c->stream);
        c->release(p);
    }
};
}  // namespace

extern "C" {

int vpbs_ctx_create(int device_ordinal, unsigned log_n_max, unsigned rate_bits, unsigned cap_height, vpbs_ctx** out) {
    if (!out || log_n_max == 0 || log_n_max + rate_bits > 24 || rate_bits > 4) return VPBS_ERR_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_ordinal < 0 || device_ordinal >= count) return VPBS_ERR_DEVICE;
    if (hipSetDevice(device_ordinal) != hipSuccess) return VPBS_ERR_DEVICE;
    auto* c = new vpbs_ctx();
    c->device = device_ordinal;
    c->log_n_max = log_n_max;
    c->rate_bits = rate_bits;
    c->cap_height = cap_height;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) != hipSuccess) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return VPBS_ERR_DEVICE;
    }
    *out = c;
    return VPBS_OK;
}

void vpbs_ctx_destroy(vpbs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)vpbs::stream_sync(c->stream);
    if (c->d_clock_samples) c->release(c->d_clock_samples);
    for (auto& kv : c->root_tables) c->release(kv.second);
    for (auto& kv : c->prescale_tables) c->release(kv.second);
    for (auto& kv : c->lde_tables) c->release(kv.second);
    for (auto& kv : c->l0_tables) c->release(kv.second);
    for (auto& kv : c->ring_tables) c->release(kv.second);
    c->resolve_timing();
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto& kv : c->free_blocks) (void)hipFree(kv.second);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (auto st : c->gate_streams)
        if (st) {
            vpbs::stream_sync_forget(st);
            (void)hipStreamDestroy(st);
        }
    if (c->gate_fork) (void)hipEventDestroy(c->gate_fork);
    for (auto e : c->gate_join)
        if (e) (void)hipEventDestroy(e);
    (void)vpbs::stream_sync(c->upload_stream);
    vpbs::stream_sync_forget(c->upload_stream);
    vpbs::stream_sync_forget(c->stream);
    (void)hipStreamDestroy(c->upload_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int vpbs_ctx_set_gate_lanes(vpbs_ctx* c, unsigned lanes) {
    if (!c || (lanes != 1 && lanes != 3)) return VPBS_ERR_INVALID;
    c->gate_lanes = lanes;
    return VPBS_OK;
}
int vpbs_ctx_set_option(vpbs_ctx* c, int option, uint64_t value) {
    if (!c) return VPBS_ERR_INVALID;
    switch (option) {
        case VPBS_OPT_GATE_LANES: return vpbs_ctx_set_gate_lanes(c, (unsigned)value);
        case VPBS_OPT_GATES_FUSED: c->tune.gates_fused = value != 0; return VPBS_OK;
        case VPBS_OPT_GATE_ITEMS:
            if (value < 1 || value > 8) return VPBS_ERR_INVALID;
            c->tune.gate_items = (unsigned)value;
            return VPBS_OK;
        case VPBS_OPT_WIDE_THRESHOLD: c->tune.wide_threshold = c->tune.fri_leaf_wide_threshold = (size_t)value; return VPBS_OK;   // as the environment variable
        case VPBS_OPT_MERKLE_CLIMB: c->tune.merkle_climb = value != 0; return VPBS_OK;
        case VPBS_OPT_GATES_TILE: c->tune.gates_tile = value != 0; return VPBS_OK;
        default: return VPBS_ERR_INVALID;
    }
}
int vpbs_ctx_get_option(const vpbs_ctx* c, int option, uint64_t* out) {
    if (!c || !out) return VPBS_ERR_INVALID;
    switch (option) {
        case VPBS_OPT_GATE_LANES: *out = c->gate_lanes; return VPBS_OK;
        case VPBS_OPT_GATES_FUSED: *out = c->tune.gates_fused; return VPBS_OK;
        case VPBS_OPT_GATE_ITEMS: *out = c->tune.gate_items; return VPBS_OK;
        case VPBS_OPT_WIDE_THRESHOLD: *out = c->tune.wide_threshold; return VPBS_OK;
        case VPBS_OPT_MERKLE_CLIMB: *out = c->tune.merkle_climb; return VPBS_OK;
        case VPBS_OPT_GATES_TILE: *out = c->tune.gates_tile; return VPBS_OK;
        default: return VPBS_ERR_INVALID;
    }
}
const char* vpbs_last_error(const vpbs_ctx* c) { return c ? c->err.c_str() : "null context"; }
int vpbs_ctx_synchronize(vpbs_ctx* c) {
    return guarded(c, [&] { VPBS_HIP(vpbs::stream_sync(c->stream)); });
}
void* vpbs_ctx_stream(vpbs_ctx* c) { return c ? (void*)c->stream : nullptr; }
// the switch table (include/vpbs_prover.h): defaults = plonky2 0.2.0 as restated
void vpbs_compat_default(vpbs_compat* out) {
    if (out) *out = vpbs_compat{0, 1, 1, 1};
}
static bool compat_supported(const vpbs_compat& k) {
    return (k.fri_mul_final_by_x == 0 || k.fri_mul_final_by_x == 1) && (k.bytes_pi_len_prefix == 0 || k.bytes_pi_len_prefix == 1) &&
           (k.digest_domain_separator == 0 || k.digest_domain_separator == 1) && k.pow_smallest_nonce == 1;
}
int vpbs_ctx_set_compat(vpbs_ctx* c, const vpbs_compat* compat) {
    if (!c || !compat || !compat_supported(*compat)) return VPBS_ERR_INVALID;
    c->compat = *compat;
    return VPBS_OK;
}
int vpbs_ctx_get_compat(const vpbs_ctx* c, vpbs_compat* out) {
    if (!c || !out) return VPBS_ERR_INVALID;
    *out = c->compat;
    return VPBS_OK;
}
unsigned vpbs_ctx_rate_bits(const vpbs_ctx* c) { return c ? c->rate_bits : 0; }
unsigned vpbs_ctx_cap_height(const vpbs_ctx* c) { return c ? c->cap_height : 0; }
int vpbs_ctx_device(const vpbs_ctx* c) { return c ? c->device : -1; }

// shader clock of one CU over ~20 us: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz
__global__ void clock_probe_kernel(unsigned long long* out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned x = threadIdx.x + 1;
    unsigned long long r1 = r0;
    while (r1 - r0 < 2000) {   // 20 us
        for (int k = 0; k < 64; ++k) x = x * 2654435761u + 12345u;
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
        out[2] = x;
    }
}
int vpbs_k_clock_probe(vpbs_ctx* c, double* mhz_out) {
    if (!c || !mhz_out) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        u64* d = c->alloc_words(4);
        clock_probe_kernel<<<1, 64, 0, c->stream>>>(reinterpret_cast<unsigned long long*>(d));
        u64 h[3] = {0, 0, 0};
        c->d2h_sync(h, d, sizeof h);
        c->release(d);
        *mhz_out = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    });
}

static int commit_any(vpbs_ctx* c, const u64* data, bool on_device, bool is_values, unsigned ncols, unsigned log_n, vpbs_batch** out,
                      u64* cap_out) {
    if (!c || !data || !out) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        VPBS_HIP(hipSetDevice(c->device));
        vpbs_batch* b = nullptr;
        if (on_device) {
            b = vpbs::commit_device(c, data, ncols, log_n, is_values);
        } else {
            VPBS_REQUIRE(log_n <= c->log_n_max, "log_n exceeds the context's log_n_max");
            DevTemp tmp(c, data, (size_t)ncols << log_n);
            b = vpbs::commit_device(c, tmp.p, ncols, log_n, is_values);
        }
        if (cap_out) {
            try {
                vpbs::batch_cap_to_host(b, cap_out);
            } catch (...) {
                vpbs_batch_free(b);
                throw;
            }
        }
        *out = b;
    });
}
int vpbs_commit_values(vpbs_ctx* c, const uint64_t* v, unsigned ncols, unsigned log_n, vpbs_batch** out, uint64_t* cap) {
    return commit_any(c, v, false, true, ncols, log_n, out, cap);
}
int vpbs_commit_coeffs(vpbs_ctx* c, const uint64_t* v, unsigned ncols, unsigned log_n, vpbs_batch** out, uint64_t* cap) {
    return commit_any(c, v, false, false, ncols, log_n, out, cap);
}
int vpbs_commit_values_dev(vpbs_ctx* c, const uint64_t* v, unsigned ncols, unsigned log_n, vpbs_batch** out, uint64_t* cap) {
    return commit_any(c, v, true, true, ncols, log_n, out, cap);
}
int vpbs_commit_coeffs_dev(vpbs_ctx* c, const uint64_t* v, unsigned ncols, unsigned log_n, vpbs_batch** out, uint64_t* cap) {
    return commit_any(c, v, true, false, ncols, log_n, out, cap);
}

int vpbs_commit_sharded_dev(vpbs_ctx* c, const uint64_t* d_data, int is_values, unsigned ncols, unsigned log_n, unsigned shard,
                            unsigned n_shards, vpbs_batch** out, uint64_t* local_cap_out) {
    if (!c || !d_data || !out) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        vpbs_batch* b = vpbs::commit_device(c, d_data, ncols, log_n, is_values != 0, shard, n_shards);
        if (local_cap_out) {
            try {
                vpbs::batch_cap_to_host(b, local_cap_out);
            } catch (...) {
                vpbs_batch_free(b);
                throw;
            }
        }
        *out = b;
    });
}

void vpbs_batch_free(vpbs_batch* b) {
    if (!b) return;
    // stream-ordered reuse: later work on the same stream may take these blocks; nothing else touches them
    b->ctx->release(b->d_coeffs);
    b->ctx->release(b->d_lde);
    b->ctx->release(b->d_digests);
    delete b;
}
unsigned vpbs_batch_ncols(const vpbs_batch* b) { return b ? b->ncols : 0; }
unsigned vpbs_batch_log_n(const vpbs_batch* b) { return b ? b->log_n : 0; }

int vpbs_batch_cap(vpbs_batch* b, uint64_t* cap_out) {
    if (!b || !cap_out) return VPBS_ERR_INVALID;
    return guarded(b->ctx, [&] { vpbs::batch_cap_to_host(b, cap_out); });
}
int vpbs_batch_coeffs(vpbs_batch* b, uint64_t* out) {
    if (!b || !out) return VPBS_ERR_INVALID;
    return guarded(b->ctx, [&] {
        VPBS_HIP(hipMemcpyAsync(out, b->d_coeffs, sizeof(u64) * b->ncols * b->n(), hipMemcpyDeviceToHost, b->ctx->stream));
        VPBS_HIP(vpbs::stream_sync(b->ctx->stream));
    });
}

int vpbs_batch_lde_rows(vpbs_batch* b, size_t row_start, size_t nrows, size_t step, uint64_t* out) {
    if (!b || !out) return VPBS_ERR_INVALID;
    return guarded(b->ctx, [&] {
        vpbs_ctx* c = b->ctx;
        VPBS_REQUIRE(b->n_shards == 1, "get_lde_values is not available on a sharded batch");
        const size_t L = b->lde_len();
        const unsigned log_L = b->log_n + c->rate_bits;
        VPBS_REQUIRE(nrows <= vpbs::MAX_QUERIES * (size_t)4096, "too many rows in one call");
        // reuse the query-open kernel: one "tree" without siblings, chunks of MAX_QUERIES rows
        u64* d_out = c->alloc_words(nrows * b->ncols);
        vpbs::OpenArgs* d_args = static_cast<vpbs::OpenArgs*>(c->alloc_bytes(sizeof(vpbs::OpenArgs)));
        for (size_t done = 0; done < nrows; done += vpbs::MAX_QUERIES) {
            vpbs::OpenArgs a{};
            const unsigned cnt = (unsigned)std::min<size_t>(vpbs::MAX_QUERIES, nrows - done);
            a.n_trees = 1;
            a.n_queries = cnt;
            a.record_words = b->ncols;
            a.trees[0].data0 = b->d_lde;
            a.trees[0].col_stride = L;
            a.trees[0].leaf_len = b->ncols;
            a.trees[0].leaf_lo = 0;
            a.trees[0].leaf_hi = L;
            for (unsigned q = 0; q < cnt; ++q) {
                const size_t idx = (row_start + done + q) * step;
                VPBS_REQUIRE(idx < L, "LDE row out of range");
                a.x_index[q] = gl::bitrev32((vpbs::u32)idx, log_L);
            }
            VPBS_HIP(hipMemcpyAsync(d_args, &a, sizeof a, hipMemcpyHostToDevice, c->stream));
            vpbs::launch_open_queries(c->stream, d_args, 1, cnt, d_out + done * b->ncols);
            VPBS_HIP(vpbs::stream_sync(c->stream));
        }
        VPBS_HIP(hipMemcpy(out, d_out, sizeof(u64) * nrows * b->ncols, hipMemcpyDeviceToHost));
        c->release(d_out);
        c->release(d_args);
    });
}

int vpbs_batch_eval_ext(vpbs_batch* b, const uint64_t zeta[2], uint64_t* out) {
    if (!b || !zeta || !out) return VPBS_ERR_INVALID;
    return guarded(b->ctx, [&] {
        vpbs_ctx* c = b->ctx;
        const size_t n = b->n();
        const unsigned chunks = (unsigned)((n + 4095) / 4096);
        u64* zpow = c->alloc_words(2 * n);
        u64* d_out = c->alloc_words(2 * (size_t)b->ncols * (1 + chunks));
        const gl::Ext zp{zeta[0], zeta[1]};
        vpbs::launch_ext_powers(c->stream, &zp, 1, n, zpow);
        vpbs::launch_eval_ext(c->stream, b->d_coeffs, b->ncols, n, n, zpow, d_out);
        VPBS_HIP(hipMemcpyAsync(out, d_out, sizeof(u64) * 2 * b->ncols, hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
        c->release(zpow);
        c->release(d_out);
    });
}

int vpbs_batch_open(vpbs_batch* b, size_t leaf_index, uint64_t* leaf_out, uint64_t* siblings_out) {
    if (!b || !leaf_out || !siblings_out) return VPBS_ERR_INVALID;
    return guarded(b->ctx, [&] {
        vpbs_ctx* c = b->ctx;
        VPBS_REQUIRE(leaf_index >= b->leaf_offset() && leaf_index < b->leaf_offset() + b->lde_len(),
                     "leaf index outside this batch's shard");
        leaf_index -= b->leaf_offset();
        vpbs::OpenArgs a{};
        a.n_trees = 1;
        a.n_queries = 1;
        vpbs::OpenTree& t = a.trees[0];
        t.data0 = b->d_lde;
        t.digests = b->d_digests;
        t.col_stride = b->lde_len();
        t.leaf_len = b->ncols;
        t.n_siblings = b->n_levels() - 1;
        t.leaf_lo = 0;
        t.leaf_hi = b->lde_len();
        for (unsigned k = 0; k < b->n_levels(); ++k) t.level_off[k] = b->level_off[k];
        a.record_words = b->ncols + 4 * (size_t)t.n_siblings;
        a.x_index[0] = leaf_index;
        u64* d_out = c->alloc_words(a.record_words);
        auto* d_args = static_cast<vpbs::OpenArgs*>(c->alloc_bytes(sizeof a));
        VPBS_HIP(hipMemcpyAsync(d_args, &a, sizeof a, hipMemcpyHostToDevice, c->stream));
        vpbs::launch_open_queries(c->stream, d_args, 1, 1, d_out);
        std::vector<u64> rec(a.record_words);
        VPBS_HIP(hipMemcpyAsync(rec.data(), d_out, sizeof(u64) * a.record_words, hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
        std::memcpy(leaf_out, rec.data(), sizeof(u64) * b->ncols);
        std::memcpy(siblings_out, rec.data() + b->ncols, sizeof(u64) * 4 * t.n_siblings);
        c->release(d_out);
        c->release(d_args);
    });
}

// ---------------- kernel-level hooks ----------------
int vpbs_k_poseidon_batch(vpbs_ctx* c, uint64_t* states, size_t n) {
    if (!c || !states) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        DevTemp d(c, states, 12 * n);
        vpbs::launch_permute_batch(c->stream, d.p, n);
        VPBS_HIP(hipMemcpyAsync(states, d.p, sizeof(u64) * 12 * n, hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}
int vpbs_k_hash_rows(vpbs_ctx* c, const uint64_t* rows, size_t n, unsigned len, uint64_t* out) {
    if (!c || !rows || !out || len == 0) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        DevTemp d(c, rows, n * len);
        DevTemp o(c, nullptr, 4 * n);
        vpbs::launch_hash_rows(c->stream, d.p, n, len, o.p);
        VPBS_HIP(hipMemcpyAsync(out, o.p, sizeof(u64) * 4 * n, hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}
int vpbs_k_intt(vpbs_ctx* c, const uint64_t* values, unsigned ncols, unsigned log_n, uint64_t* coeffs_out) {
    if (!c || !values || !coeffs_out) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        const size_t words = (size_t)ncols << log_n;
        DevTemp in(c, values, words), out(c, nullptr, words), scratch(c, nullptr, words);
        vpbs::launch_intt(c->stream, in.p, out.p, scratch.p, c->roots(log_n, true), ncols, log_n);
        VPBS_HIP(hipMemcpyAsync(coeffs_out, out.p, sizeof(u64) * words, hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}
int vpbs_k_coset_lde(vpbs_ctx* c, const uint64_t* coeffs, unsigned ncols, unsigned log_n, unsigned rate_bits, uint64_t shift,
                     uint64_t* out_host) {
    if (!c || !coeffs || !out_host) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        const size_t words = (size_t)ncols << log_n;
        DevTemp in(c, coeffs, words), out(c, nullptr, words << rate_bits);
        vpbs::launch_coset_lde(c->stream, in.p, out.p, c->roots(log_n, false), c->lde_table(log_n, rate_bits, shift), ncols, log_n,
                               rate_bits);
        VPBS_HIP(hipMemcpyAsync(out_host, out.p, sizeof(u64) * (words << rate_bits), hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}
int vpbs_k_merkle_cap(vpbs_ctx* c, const uint64_t* leaves, size_t n_leaves, unsigned leaf_len, unsigned cap_height, uint64_t* cap_out) {
    if (!c || !leaves || !cap_out || leaf_len == 0) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        std::vector<size_t> off;
        const size_t words = vpbs::merkle_layout(n_leaves, cap_height, off);
        DevTemp in(c, leaves, n_leaves * leaf_len), dig(c, nullptr, words);
        if (leaf_len <= 4) {
            // hash_or_noop: padded copy
            std::vector<u64> padded(4 * n_leaves, 0);
            for (size_t i = 0; i < n_leaves; ++i)
                for (unsigned k = 0; k < leaf_len; ++k) padded[4 * i + k] = leaves[i * leaf_len + k];
            VPBS_HIP(hipMemcpyAsync(dig.p, padded.data(), sizeof(u64) * 4 * n_leaves, hipMemcpyHostToDevice, c->stream));
            VPBS_HIP(vpbs::stream_sync(c->stream));
        } else {
            vpbs::launch_hash_rows(c->stream, in.p, n_leaves, leaf_len, dig.p);
        }
        vpbs::launch_merkle_tree(c->stream, c->tune, dig.p, off.data(), (unsigned)off.size(), n_leaves);
        VPBS_HIP(hipMemcpyAsync(cap_out, dig.p + off.back(), sizeof(u64) * ((size_t)4 << cap_height), hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}

int vpbs_ntt_params(unsigned log_n, uint64_t* roots, uint64_t* invroots, uint64_t* ninv) {
    if (!roots || !invroots || !ninv || log_n == 0 || log_n > 16) return VPBS_ERR_INVALID;
    // gen_param_file.sage: psi = 7^((p-1)/2N); ROOTS[j] = psi^bitrev(j); INVROOTS[j] = psi^-bitrev(j); NINV = N^-1
    const size_t n = (size_t)1 << log_n;
    const u64 psi = gl::pow(gl::GENERATOR, (gl::P - 1) / (2 * n)), psi_inv = gl::inv(psi);
    for (size_t j = 0; j < n; ++j) {
        const u64 e = gl::bitrev32((vpbs::u32)j, log_n);
        roots[j] = gl::pow(psi, e);
        invroots[j] = gl::pow(psi_inv, e);
    }
    *ninv = gl::inv((u64)n);
    return VPBS_OK;
}
int vpbs_k_negacyclic_ntt(vpbs_ctx* c, uint64_t* data, unsigned batch, unsigned log_n, int inverse) {
    if (!c || !data || log_n == 0 || log_n > 11) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        const size_t n = (size_t)1 << log_n;
        std::vector<u64> roots(n), inv(n);
        u64 ninv;
        vpbs_ntt_params(log_n, roots.data(), inv.data(), &ninv);
        DevTemp tab(c, inverse ? inv.data() : roots.data(), n), d(c, data, batch * n);
        vpbs::launch_negacyclic(c->stream, d.p, tab.p, batch, log_n, inverse != 0, ninv);
        VPBS_HIP(hipMemcpyAsync(data, d.p, sizeof(u64) * batch * n, hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}

int vpbs_blind_rotate_step(vpbs_ctx* c, const vpbs_tfhe_params* prm, unsigned batch, const uint64_t* acc_in, const uint64_t* masks,
                           const uint64_t* ggsw, int ggsw_per_instance, int first_step, int last_step, uint64_t* acc_out, int on_device) {
    if (!c || !prm || !acc_in || !masks || !acc_out || batch == 0 || (!first_step && !ggsw)) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        const unsigned log_n = prm->log_N, K = prm->K, ELL = prm->ELL, LOGB = prm->LOGB;
        VPBS_REQUIRE(log_n >= 1 && log_n <= 11 && K >= 1 && K <= 8 && LOGB >= 1 && LOGB <= 32, "unsupported TFHE parameters");
        const unsigned nl = (64 + LOGB - 1) / LOGB;
        VPBS_REQUIRE(ELL >= 1 && ELL <= nl, "ELL exceeds the number of limbs");
        const size_t n = (size_t)1 << log_n;
        VPBS_REQUIRE(ELL * n * sizeof(u64) <= 128 * 1024, "ELL * N does not fit the LDS budget");
        VPBS_REQUIRE(!(first_step && last_step), "a step cannot be both the first and the last");
        const size_t acc_words = (size_t)batch * K * n, ggsw_words = (size_t)K * ELL * K * n * (ggsw_per_instance ? batch : 1);
        const u64* tab = c->ring_table(log_n);
        u64 ninv = gl::inv((u64)n);
        std::vector<void*> tmp;
        struct Cleanup {
            vpbs_ctx* c;
            std::vector<void*>& v;
            ~Cleanup() {
                (void)vpbs::stream_sync(c->stream);
                for (void* p : v) c->release(p);
            }
        } cleanup{c, tmp};
        auto stage = [&](const u64* host, size_t words) -> u64* {
            u64* d = c->alloc_words(words);
            tmp.push_back(d);
            if (host) VPBS_HIP(hipMemcpyAsync(d, host, sizeof(u64) * words, hipMemcpyHostToDevice, c->stream));
            return d;
        };
        const u64 *d_acc = acc_in, *d_masks = masks, *d_ggsw = ggsw;
        u64* d_out = acc_out;
        if (!on_device) {
            d_acc = stage(acc_in, acc_words);
            d_masks = stage(masks, batch);
            d_ggsw = first_step ? nullptr : stage(ggsw, ggsw_words);
            d_out = stage(nullptr, acc_words);
        }
        u64* limbs = stage(nullptr, (size_t)batch * K * ELL * n);
        {
            vpbs::Timed t(c, "blind_rotate_step");
            vpbs::launch_blind_rotate_step(c->stream, d_acc, d_masks, d_ggsw, ggsw_per_instance ? (size_t)K * ELL * K * n : 0, tab, tab + n, ninv,
                                           log_n, K, ELL, LOGB, batch, first_step, last_step, limbs, d_out);
        }
        VPBS_HIP(hipGetLastError());
        if (!on_device) {
            VPBS_HIP(hipMemcpyAsync(acc_out, d_out, sizeof(u64) * acc_words, hipMemcpyDeviceToHost, c->stream));
            VPBS_HIP(vpbs::stream_sync(c->stream));
        }
    });
}

int vpbs_pbs_accumulator_chain(vpbs_ctx* c, const vpbs_tfhe_params* prm, unsigned n_lwe, const uint64_t* acc_init, const uint64_t* lwe_ct,
                               const uint64_t* bsk, const uint64_t* ksk, uint64_t* accs_out) {
    if (!c || !prm || !acc_init || !lwe_ct || !bsk || !ksk || !accs_out || n_lwe == 0) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        const unsigned log_n = prm->log_N, K = prm->K, ELL = prm->ELL, LOGB = prm->LOGB;
        VPBS_REQUIRE(log_n >= 1 && log_n <= 11 && K >= 1 && K <= 8 && LOGB >= 1 && LOGB <= 32, "unsupported TFHE parameters");
        VPBS_REQUIRE(ELL >= 1 && ELL <= (64 + LOGB - 1) / LOGB, "ELL exceeds the number of limbs");
        const size_t n = (size_t)1 << log_n;
        VPBS_REQUIRE(ELL * n * sizeof(u64) <= 128 * 1024, "ELL * N does not fit the LDS budget");
        const size_t acc_words = (size_t)K * n, ggsw_words = (size_t)K * ELL * K * n;
        const u64* tab = c->ring_table(log_n);
        const u64 ninv = gl::inv((u64)n);
        // device: every accumulator of the chain (so each step reads its predecessor in place), masks in step order, keys
        DevTemp accs(c, nullptr, (size_t)(n_lwe + 3) * acc_words), keys(c, bsk, (size_t)n_lwe * ggsw_words), kk(c, ksk, ggsw_words);
        std::vector<u64> h_masks(n_lwe + 2);
        h_masks[0] = lwe_ct[n_lwe];
        for (unsigned x = 0; x < n_lwe; ++x) h_masks[1 + x] = lwe_ct[x];
        h_masks[n_lwe + 1] = 0;
        DevTemp masks(c, h_masks.data(), h_masks.size()), limbs(c, nullptr, (size_t)K * ELL * n);
        VPBS_HIP(hipMemcpyAsync(accs.p, acc_init, sizeof(u64) * acc_words, hipMemcpyHostToDevice, c->stream));
        {
            vpbs::Timed t(c, "pbs_accumulator_chain");
            for (unsigned step = 0; step < n_lwe + 2; ++step) {
                const bool first = step == 0, last = step == n_lwe + 1;
                const u64* ggsw = first ? nullptr : (last ? kk.p : keys.p + (size_t)(step - 1) * ggsw_words);
                vpbs::launch_blind_rotate_step(c->stream, accs.p + (size_t)step * acc_words, masks.p + step, ggsw, 0, tab, tab + n, ninv, log_n, K,
                                               ELL, LOGB, 1, first, last, limbs.p, accs.p + (size_t)(step + 1) * acc_words);
            }
        }
        VPBS_HIP(hipGetLastError());
        VPBS_HIP(hipMemcpyAsync(accs_out, accs.p + acc_words, sizeof(u64) * (size_t)(n_lwe + 2) * acc_words, hipMemcpyDeviceToHost, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}

// ---------------- memory helpers for hosts without the HIP runtime ----------------
void* vpbs_host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
void vpbs_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}
int vpbs_device_alloc(vpbs_ctx* c, size_t words, uint64_t** out) {
    if (!c || !out || !words) return VPBS_ERR_INVALID;
    return guarded(c, [&] { *out = c->alloc_words(words); });
}
int vpbs_device_upload(vpbs_ctx* c, uint64_t* d_dst, const uint64_t* host_src, size_t words) {
    if (!c || !d_dst || !host_src) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        VPBS_HIP(hipMemcpyAsync(d_dst, host_src, sizeof(u64) * words, hipMemcpyHostToDevice, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}
int vpbs_device_upload_bg(vpbs_ctx* c, uint64_t* d_dst, const uint64_t* host_src, size_t words) {
    // nothing of the context is touched except its (immutable) device ordinal and upload stream: safe beside a running prover call
    if (!c || !d_dst || !host_src) return VPBS_ERR_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return VPBS_ERR_DEVICE;
    if (hipMemcpyAsync(d_dst, host_src, sizeof(u64) * words, hipMemcpyHostToDevice, c->upload_stream) != hipSuccess) return VPBS_ERR_DEVICE;
    return vpbs::stream_sync(c->upload_stream) == hipSuccess ? VPBS_OK : VPBS_ERR_DEVICE;
}
int vpbs_device_upload_rows(vpbs_ctx* c, uint64_t* d_dst, const uint64_t* host_src, unsigned n_cols, size_t n, size_t row_lo, size_t row_hi) {
    if (!c || !d_dst || !host_src || row_lo > row_hi || row_hi > n) return VPBS_ERR_INVALID;
    if (row_lo == row_hi || n_cols == 0) return VPBS_OK;
    return guarded(c, [&] {
        VPBS_HIP(hipMemcpy2DAsync(d_dst + row_lo, n * sizeof(u64), host_src + row_lo, n * sizeof(u64), (row_hi - row_lo) * sizeof(u64), n_cols,
                                  hipMemcpyHostToDevice, c->stream));
        VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}
namespace {
__global__ void __launch_bounds__(256) scatter_words_kernel(vpbs::u64* __restrict__ dst, const uint32_t* __restrict__ pos,
                                                            const vpbs::u64* __restrict__ val, size_t count) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < count) dst[pos[i]] = val[i];
}
}  // namespace
}  // extern "C"
namespace vpbs {
// the copy + scatter of vpbs_device_scatter QUEUED on the context's stream; wait = false returns at once: everything later on that stream
// (the step proof) is ordered behind it, and the caller leaves host_values / d_stage alone until it has waited on the context for
// something queued afterwards.  The IVC driver's form: with several chains per GPU the wait was 2-5 ms of standing in the device's queue
// behind the other chains' kernels, on every chain's critical path, for a result nobody reads on the host.
int device_scatter(vpbs_ctx* c, uint64_t* d_dst, const uint64_t* d_positions, const uint64_t* host_values, size_t count, uint64_t* d_stage, bool wait) {
    if (!c || !d_dst || !d_positions || !host_values || !d_stage) return VPBS_ERR_INVALID;
    if (count == 0) return VPBS_OK;
    return guarded(c, [&] {
        VPBS_HIP(hipMemcpyAsync(d_stage, host_values, sizeof(u64) * count, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(scatter_words_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, d_dst,
                           reinterpret_cast<const uint32_t*>(d_positions), d_stage, count);
        VPBS_HIP(hipGetLastError());
        if (wait) VPBS_HIP(vpbs::stream_sync(c->stream));
    });
}
}  // namespace vpbs
extern "C" {
int vpbs_device_scatter(vpbs_ctx* c, uint64_t* d_dst, const uint64_t* d_positions, const uint64_t* host_values, size_t count,
                        uint64_t* d_stage) {
    return vpbs::device_scatter(c, d_dst, d_positions, host_values, count, d_stage, true);
}
void vpbs_device_free(vpbs_ctx* c, uint64_t* d_ptr) {
    if (c && d_ptr) {
        (void)vpbs::stream_sync(c->stream);
        c->release(d_ptr);
    }
}

// ---------------- timing ----------------
int vpbs_timing_enable(vpbs_ctx* c, int on) {
    if (!c) return VPBS_ERR_INVALID;
    c->resolve_timing();
    c->timing = on != 0;
    c->timing_only = on == 2 ? "leaf_hash" : "";  // 2: dominant kernel only (bench.py timed region)
    return VPBS_OK;
}
vpbs::u64* vpbs_ctx::next_clock_sample() {
    if (!timing) return nullptr;
    if (!d_clock_samples) {
        d_clock_samples = alloc_words(2 * CLOCK_SAMPLES);
        VPBS_HIP(hipMemsetAsync(d_clock_samples, 0, 2 * CLOCK_SAMPLES * sizeof(vpbs::u64), stream));
    }
    return d_clock_samples + 2 * (clock_samples++ % CLOCK_SAMPLES);
}
int vpbs_timing_shader_clock(vpbs_ctx* c, double* mhz_out, unsigned* samples_out) {
    if (!c || !mhz_out) return VPBS_ERR_INVALID;
    return guarded(c, [&] {
        *mhz_out = 0.0;
        const unsigned n = std::min(c->clock_samples, vpbs_ctx::CLOCK_SAMPLES);
        if (samples_out) *samples_out = n;
        if (!n || !c->d_clock_samples) return;
        std::vector<u64> h(2 * n);
        c->d2h_sync(h.data(), c->d_clock_samples, h.size() * sizeof(u64));
        double cycles = 0, ticks = 0;
        for (unsigned i = 0; i < n; ++i) {
            cycles += (double)h[2 * i];
            ticks += (double)h[2 * i + 1];
        }
        if (ticks > 0) *mhz_out = 100.0 * cycles / ticks;
        c->clock_samples = 0;
    });
}
int vpbs_timing_report(vpbs_ctx* c, char* buf, size_t len) {
    if (!c || !buf || len < 4) return VPBS_ERR_INVALID;
    c->resolve_timing();
    std::string s = "{";
    bool first = true;
    for (auto& kv : c->totals) {
        char tmp[256];
        std::snprintf(tmp, sizeof tmp, "%s\"%s\": {\"ms\": %.6f, \"count\": %ld}", first ? "" : ", ", kv.first.c_str(), kv.second.first,
                      kv.second.second);
        s += tmp;
        first = false;
    }
    s += "}";
    c->totals.clear();
    if (s.size() + 1 > len) return VPBS_ERR_INVALID;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return VPBS_OK;
}

// ---------------- host-side Challenger / hashing (iop/challenger.rs, hash/hashing.rs) ----------------
void vpbs_challenger_init(vpbs_challenger_state* ch) { std::memset(ch, 0, sizeof *ch); }
static void duplexing(vpbs_challenger_state* ch) {
    for (uint32_t i = 0; i < ch->input_len; ++i) ch->sponge[i] = ch->input[i];  // overwrite mode
    ch->input_len = 0;
    poseidon::permute_host(ch->sponge);
    for (int i = 0; i < 8; ++i) ch->output[i] = ch->sponge[i];
    ch->output_len = 8;
}
void vpbs_challenger_observe(vpbs_challenger_state* ch, const uint64_t* e, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        ch->output_len = 0;
        ch->input[ch->input_len++] = e[i];
        if (ch->input_len == 8) duplexing(ch);
    }
}
uint64_t vpbs_challenger_get(vpbs_challenger_state* ch) {
    if (ch->input_len != 0 || ch->output_len == 0) duplexing(ch);
    return ch->output[--ch->output_len];
}
void vpbs_hash_no_pad(const uint64_t* in, size_t n, uint64_t out[4]) { poseidon::hash_no_pad_host(in, n, out); }
// Hasher::hash_pad (plonk/config.rs): pad10*1 -- push 1, zeros until len + 1 is a multiple of the rate, push 1 -- then hash_no_pad
void vpbs_hash_pad(const uint64_t* in, size_t n, uint64_t out[4]) {
    std::vector<uint64_t> padded(in, in + (in ? n : 0));
    padded.push_back(1);
    while ((padded.size() + 1) % 8 != 0) padded.push_back(0);
    padded.push_back(1);
    poseidon::hash_no_pad_host(padded.data(), padded.size(), out);
}
// CircuitBuilder::build (plonk/circuit_builder.rs): circuit_digest_parts = [constants_sigmas_cap.flatten(), hash_pad(domain_separator),
// [degree_bits]] -> hash_no_pad of the concatenation.  The reference never sets a domain separator (ivc_based_vpbs.rs:190-276 build the
// circuit with CircuitBuilder::new + build), so it is the empty vector.
int vpbs_circuit_digest(const vpbs_compat* compat, const uint64_t* cap, size_t cap_words, unsigned degree_bits, uint64_t out[4]) {
    if (!cap || !out || cap_words == 0 || cap_words % 4 != 0) return VPBS_ERR_INVALID;
    vpbs_compat k;
    vpbs_compat_default(&k);
    if (compat) k = *compat;
    std::vector<uint64_t> parts(cap, cap + cap_words);
    if (k.digest_domain_separator) {
        uint64_t sep[4];
        vpbs_hash_pad(nullptr, 0, sep);
        parts.insert(parts.end(), sep, sep + 4);
    }
    parts.push_back(degree_bits);
    poseidon::hash_no_pad_host(parts.data(), parts.size(), out);
    return VPBS_OK;
}

int vpbs_host_set_poseidon_x8(int on) {
#if defined(VPBS_HAVE_POSEIDON_X8)
    poseidon_x8::switch_state().store(on ? 1 : 0);
    return poseidon_x8::enabled() ? 1 : 0;
#else
    (void)on;
    return 0;
#endif
}

int vpbs_k_poseidon_host(uint64_t* states, size_t n) {
    if (!states && n) return VPBS_ERR_INVALID;
#if defined(VPBS_HAVE_POSEIDON_X8)
    if (poseidon_x8::enabled()) {
        poseidon_x8::permute_many(states, n);
        return 1;
    }
#endif
    for (size_t i = 0; i < n; ++i) poseidon::permute_host(states + 12 * i);
    return 0;
}

int vpbs_hash_chain_links(const uint64_t prefix[4], const uint64_t* const* items, size_t n_links, size_t item_len, uint64_t* out) {
    if (!prefix || (n_links && (!items || !out))) return VPBS_ERR_INVALID;
    for (size_t k = 0; k < n_links; ++k)
        if (!items[k] && item_len) return VPBS_ERR_INVALID;
    vpbs::hash_links_shared(prefix, items, n_links, item_len, out);
    return VPBS_OK;
}

int vpbs_hash_chain(const uint64_t* items, size_t n_items, size_t item_len, const uint64_t claimed[4], uint64_t out[4]) {
    if (n_items && !items) return VPBS_ERR_INVALID;
    u64 h[4] = {0, 0, 0, 0};
    for (size_t k = 0; k < n_items; ++k) {
        // hash_no_pad(h || item): overwrite-mode sponge over the 4 + item_len elements, without materialising the concatenation
        const u64* item = items + k * item_len;
        u64 next[4];
        vpbs::hash_links_shared(h, &item, 1, item_len, next);
        std::memcpy(h, next, sizeof h);
    }
    if (out) std::memcpy(out, h, sizeof h);
    if (!claimed) return 1;
    return std::memcmp(h, claimed, sizeof h) == 0 ? 1 : 0;
}

}  // extern "C"
