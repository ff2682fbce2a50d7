/* TEST INFRASTRUCTURE -- not part of the product, never linked into it.
 *
 * A stand-in for librccl.so that lets the library's NATIVE communicator (csrc/comm_rccl.hip: vpbs_comm_rccl_create, its stream polling,
 * its timeout + ncclCommAbort path, the [world + 1][SMALL_WORDS] staging, the pinned device-to-host path, the status words of a sharded
 * step) run with MORE THAN ONE RANK on a box with ONE GPU: the seven entry points comm_rccl.hip binds, with the ranks = processes that
 * share the device and exchange through a POSIX shared-memory segment.  Loaded only when a test sets VPBS_RCCL_LIB to this file's .so.
 *
 * A collective is stream-ordered like the real one: the send buffer is copied to pinned host memory ON THE CALLER'S STREAM, a host
 * function on that stream puts it into the rank's slot of the segment, meets the other ranks at a barrier, gathers (or sums) all slots
 * into pinned memory and meets them again (the slots are free for the next collective), and the result is copied to the receive buffer
 * on the same stream.  A peer that never arrives leaves the host function waiting -- the stream stays busy, which is exactly what a hung
 * ncclAllGather looks like to comm_rccl.hip's wait_collective; ncclCommAbort releases it.
 *
 * Build: gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.c -o tests/_build/libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#define FAKE_MAGIC 0x4b4146454c434352ull /* "RCCLEFAK" */

typedef struct {
    _Atomic uint64_t magic;
    _Atomic uint32_t attached;    /* ranks that have mapped the segment */
    _Atomic uint32_t arrived;     /* barrier: arrivals of the current generation */
    _Atomic uint32_t generation;  /* barrier: bumped by the last arrival */
    _Atomic uint32_t destroyed;   /* ranks that called ncclCommDestroy */
    uint64_t slot_bytes;
    uint32_t nranks;
} fake_header;

struct ncclComm {
    int rank, nranks;
    size_t slot_bytes, map_bytes;
    char name[64];
    fake_header* hdr;
    uint8_t* slots;               /* [nranks][slot_bytes] in the shared segment */
    uint8_t* pin_local;           /* pinned: this rank's contribution */
    uint8_t* pin_full;            /* pinned: [nranks][bytes] gathered, or the reduced vector */
    _Atomic int aborted;
    _Atomic uint64_t collectives; /* statistics for the tests: collectives completed on this rank */
};

typedef struct {
    struct ncclComm* c;
    size_t bytes;
    int reduce;                   /* 0 = all-gather, 1 = sum of uint64 */
} fake_op;

static size_t header_bytes(void) { return 4096; }

/* returns 0, or -1 when the communicator was aborted while waiting */
static int barrier(struct ncclComm* c) {
    fake_header* h = c->hdr;
    const uint32_t gen = atomic_load(&h->generation);
    if (atomic_fetch_add(&h->arrived, 1) + 1 == (uint32_t)c->nranks) {
        atomic_store(&h->arrived, 0);
        atomic_fetch_add(&h->generation, 1);
        return 0;
    }
    for (unsigned spins = 0; atomic_load(&h->generation) == gen; ++spins) {
        if (atomic_load(&c->aborted)) return -1;
        if (spins > 2000) {
            struct timespec ts = {0, 20000};
            nanosleep(&ts, NULL);
        } else {
            sched_yield();
        }
    }
    return 0;
}

static void run_collective(void* user) {
    fake_op* op = (fake_op*)user;
    struct ncclComm* c = op->c;
    const size_t bytes = op->bytes;
    if (!atomic_load(&c->aborted)) {
        memcpy(c->slots + (size_t)c->rank * c->slot_bytes, c->pin_local, bytes);
        if (barrier(c) == 0) {
            if (op->reduce) {
                uint64_t* acc = (uint64_t*)c->pin_full;
                memset(acc, 0, bytes);
                for (int r = 0; r < c->nranks; ++r) {
                    const uint64_t* s = (const uint64_t*)(c->slots + (size_t)r * c->slot_bytes);
                    for (size_t i = 0; i < bytes / 8; ++i) acc[i] += s[i];
                }
            } else {
                for (int r = 0; r < c->nranks; ++r) memcpy(c->pin_full + (size_t)r * bytes, c->slots + (size_t)r * c->slot_bytes, bytes);
            }
            if (barrier(c) == 0) atomic_fetch_add(&c->collectives, 1);
        }
    }
    free(op);
}

static ncclResult_t collective(const void* send, void* recv, size_t bytes, int reduce, struct ncclComm* c, hipStream_t s) {
    if (!c || atomic_load(&c->aborted)) return ncclInvalidUsage;
    if (bytes > c->slot_bytes) return ncclInvalidArgument;
    fake_op* op = (fake_op*)malloc(sizeof *op);
    if (!op) return ncclSystemError;
    op->c = c;
    op->bytes = bytes;
    op->reduce = reduce;
    if (hipMemcpyAsync(c->pin_local, send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(s, run_collective, op) != hipSuccess) return ncclUnhandledCudaError;
    const size_t out = reduce ? bytes : bytes * (size_t)c->nranks;
    if (hipMemcpyAsync(recv, c->pin_full, out, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

static size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static _Atomic unsigned serial;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/vpbs_fake_rccl_%d_%u_%lx", (int)getpid(), atomic_fetch_add(&serial, 1), (unsigned long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    c->rank = rank;
    c->nranks = nranks;
    const char* e = getenv("FAKE_RCCL_SLOT_BYTES");
    c->slot_bytes = e ? (size_t)strtoull(e, NULL, 0) : ((size_t)16 << 20);
    c->map_bytes = header_bytes() + (size_t)nranks * c->slot_bytes;
    snprintf(c->name, sizeof c->name, "%.60s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { free(c); return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); free(c); return ncclSystemError; }
    void* m = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { free(c); return ncclSystemError; }
    c->hdr = (fake_header*)m;
    c->slots = (uint8_t*)m + header_bytes();
    if (rank == 0) {
        c->hdr->slot_bytes = c->slot_bytes;
        c->hdr->nranks = (uint32_t)nranks;
        atomic_store(&c->hdr->magic, FAKE_MAGIC);
    }
    /* like the real ncclCommInitRank: returns when every rank has joined (bounded: a rank that never comes fails the init) */
    const char* te = getenv("FAKE_RCCL_INIT_TIMEOUT_S");
    const double limit = te ? atof(te) : 120.0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (atomic_load(&c->hdr->magic) != FAKE_MAGIC) sched_yield();
    atomic_fetch_add(&c->hdr->attached, 1);
    while (atomic_load(&c->hdr->attached) < (uint32_t)nranks) {
        struct timespec ts = {0, 200000};
        nanosleep(&ts, NULL);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit) {
            munmap(m, c->map_bytes);
            shm_unlink(c->name);
            free(c);
            return ncclSystemError;
        }
    }
    if (hipHostMalloc((void**)&c->pin_local, c->slot_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->pin_full, c->slot_bytes * (size_t)nranks, hipHostMallocDefault) != hipSuccess) {
        munmap(m, c->map_bytes);
        free(c);
        return ncclUnhandledCudaError;
    }
    *out = c;
    return ncclSuccess;
}

static void release(struct ncclComm* c) {
    /* the name goes when the last rank leaves (or at once after an abort: the others may never come) */
    const uint32_t gone = atomic_fetch_add(&c->hdr->destroyed, 1) + 1;
    if (gone >= (uint32_t)c->nranks || atomic_load(&c->aborted)) shm_unlink(c->name);
    munmap((void*)c->hdr, c->map_bytes);
    (void)hipHostFree(c->pin_local);
    (void)hipHostFree(c->pin_full);
    free(c);
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    release(c);
    return ncclSuccess;
}

/* Releases a host function that waits for a peer; the communicator is unusable afterwards.  The memory stays mapped until the process
 * exits: a host function released by the abort may still be returning through it. */
ncclResult_t ncclCommAbort(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    atomic_store(&c->aborted, 1);
    shm_unlink(c->name);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
    const size_t w = type_bytes(t);
    if (!w) return ncclInvalidArgument;
    return collective(send, recv, count * w, 0, c, s);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    if (t != ncclUint64 && t != ncclInt64) return ncclInvalidArgument; /* all the prover uses: wrapping 64-bit sums */
    if (op != ncclSum) return ncclInvalidArgument;
    return collective(send, recv, count * 8, 1, c, s);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "fake rccl: success";
        case ncclInvalidArgument: return "fake rccl: invalid argument (message larger than FAKE_RCCL_SLOT_BYTES?)";
        case ncclInvalidUsage: return "fake rccl: communicator aborted";
        case ncclSystemError: return "fake rccl: shared-memory segment";
        default: return "fake rccl: hip error";
    }
}

/* test hook: collectives this rank has completed on the communicator */
uint64_t fake_rccl_collectives(ncclComm_t c) { return c ? atomic_load(&c->collectives) : 0; }
