// Witness generation on the device for a batch of PartialWitnesses of one circuit (vpbs_witness_device_*): the level schedule a compiled
// plan carries (witness_plan.h, built by witness.hip) replayed on the GPU with the generators of witness_gen.h.  SURVEY.md 8f-2.
#include <cstring>

#include "witness_plan.h"

// ---- device witness generation: one circuit, a batch of PartialWitnesses -----------------------------------------------------------
// The n + 2 step witnesses of a PBS are independent once the accumulator chain is known (vpbs_pbs_accumulator_chain), and they share
// one circuit: the plan's level schedule is replayed for all of them at once.  Values live in HBM as val[slot][batch] (instances
// innermost: every access of an operation is one coalesced run over the batch); an operation is a thread per instance.  The
// sequential spine of the step circuit is its bootstrapping-key hash chain (2 049 PoseidonGate rows, one level each), so a run is ~2 050 levels
// of small launches -- latency-bound, amortised over the batch; the wires of one instance are then gathered straight into the
// [n_wires][n] matrix vpbs_prove_step takes as a device input: they never cross PCIe.
namespace vpbs {
namespace {
using Plan = vpbs_witness_plan;
constexpr unsigned WT = 256;
enum DevErr : unsigned { DE_SET_TWICE = 1, DE_TOO_LARGE = 2, DE_NOT_BOOLEAN = 4, DE_DIV_ZERO = 8, DE_GATE = 16 };

// Which instances a launch works on: `n` instances starting at *first (0 when first is null) of a value array laid out for `stride`
// instances per slot.  A batch run: n = stride = the batch.  The late phase of ONE instance of a batch (vpbs_witness_device_run_late):
// n = 1, stride = the batch, *first = the instance -- read from device memory so that one captured graph serves every instance.
struct Launch {
    u32 stride, n;
    const u32* first;
    __device__ u32 instance(size_t gid) const { return (first ? *first : 0u) + (u32)(gid % n); }
};

// PLAIN = false: volatile loads -- inside the chain kernel a row reads what other lanes of the same group stored a moment ago (no stale L1 line).
// PLAIN = true (the walk of the late phase): ordinary loads; everything a level reads was stored in an earlier level, and the barrier between
// levels releases / acquires at device scope (the stores are out of the CU, the reader's L1 is invalidated).  The compiler may then request
// the operands of a long generator (a ReducingGate's 43 coefficients, an interpolation's 16 points) together instead of one round trip each.
template <bool PLAIN> struct ValsT {
    u64* v;
    unsigned* err;
    u32 batch, b;   // batch: the stride between slots
    __device__ u64 get(u32 slot) const {
        const u64* p = v + (size_t)(slot & ~Plan::CHECK) * batch + b;
        if constexpr (PLAIN) return *p;
        else return *(volatile const u64*)p;
    }
    __device__ void set(u32 slot, u64 x) const {
        if (x >= gl::P) x -= gl::P;
        u64* p = v + (size_t)(slot & ~Plan::CHECK) * batch + b;
        if (slot & Plan::CHECK) {
            if (*p != x) {
                atomicOr(err, DE_SET_TWICE);
                atomicCAS(err + 1, 0u, (slot & ~Plan::CHECK) + 1);   // the first class caught (for the message)
            }
        } else {
            *p = x;
        }
    }
};
using Vals = ValsT<false>;

// compare_pass 0: the presets that write their class; 1: the ones that find it written (a class preset twice: the cyclic circuit's own verifier
// data and the tail of the inner proof's public inputs) and compare -- in a launch of their own, after the writers
__global__ void __launch_bounds__(WT) wd_preset_kernel(u64* v, unsigned* err, const u32* slots, const u64* values, u32 n_preset, Launch L,
                                                        u32 compare_pass) {
    const size_t gid = blockIdx.x * (size_t)WT + threadIdx.x;
    if (gid >= (size_t)n_preset * L.n) return;
    const u32 slot = slots[gid / L.n];
    if (((slot & Plan::CHECK) != 0) != (compare_pass != 0)) return;
    const Vals a{v, err, L.stride, L.instance(gid)};
    a.set(slot, values[gid]);
}

__global__ void __launch_bounds__(WT) wd_const_kernel(u64* v, unsigned* err, const Plan::ConstOp* ops, u32 n_ops, Launch L) {
    const size_t gid = blockIdx.x * (size_t)WT + threadIdx.x;
    if (gid >= (size_t)n_ops * L.n) return;
    const Vals a{v, err, L.stride, L.instance(gid)};
    a.set(ops[gid / L.n].out, ops[gid / L.n].value);
}

template <class V> __device__ __forceinline__ void do_arith(const V& a, const Plan::ArithOp& op) {
    a.set(op.out, gl::add(gl::mul(gl::mul(a.get(op.x), a.get(op.y)), op.c0), gl::mul(a.get(op.z), op.c1)));
}
__global__ void __launch_bounds__(WT) wd_arith_kernel(u64* v, unsigned* err, const Plan::ArithOp* ops, u32 n_ops, Launch L) {
    const size_t gid = blockIdx.x * (size_t)WT + threadIdx.x;
    if (gid >= (size_t)n_ops * L.n) return;
    do_arith(Vals{v, err, L.stride, L.instance(gid)}, ops[gid / L.n]);
}

template <class V> __device__ __forceinline__ void do_bits(const V& a, const Plan::BitsOp& op, const u32* aux) {
    u64 x = a.get(op.in);
    const u64 mask = ((u64)1 << op.bits) - 1;
    for (u32 k = 0; k < op.n_out; ++k) {
        a.set(aux[op.out_at + k], x & mask);
        x >>= op.bits;
    }
    if (x != 0) atomicOr(a.err, DE_TOO_LARGE);
}
__global__ void __launch_bounds__(WT) wd_bits_kernel(u64* v, unsigned* err, const Plan::BitsOp* ops, const u32* aux, u32 n_ops, Launch L) {
    const size_t gid = blockIdx.x * (size_t)WT + threadIdx.x;
    if (gid >= (size_t)n_ops * L.n) return;
    do_bits(Vals{v, err, L.stride, L.instance(gid)}, ops[gid / L.n], aux);
}

// every gate generator without a special form: gen_run (the host's code) through the row's slot table, one thread per instance
template <class V> struct DevRow {
    V a;
    const u32* rs;
    __device__ u64 get(unsigned w) const { return a.get(rs[w]); }
    __device__ void set(unsigned w, u64 x) const { a.set(rs[w], x); }
    __device__ void fail(const char*) const { atomicOr(a.err, DE_GATE); }
};

struct RowTables {
    const vpbs_gate* gates;
    const u32 *row_gate, *row_off;
    const u64* consts;
    const gates::CosetTables* coset;  // [n_gates]
    u32 max_consts;
};

template <class V> __device__ __forceinline__ void do_rowop(const V& a, const Plan::RowOp& op, const RowTables& t, const u32* row_slots) {
    const u32 gi = t.row_gate[op.row];
    const vpbs_gate g = t.gates[gi];
    DevRow<V> r{a, row_slots + t.row_off[op.row]};
    gen_run(g, op.sub, t.consts + (size_t)op.row * t.max_consts, r, g.kind == VPBS_GATE_COSET_INTERPOLATION ? t.coset + gi : nullptr);
}
__global__ void __launch_bounds__(64) wd_rowop_kernel(u64* v, unsigned* err, const Plan::RowOp* ops, RowTables t, const u32* row_slots, u32 n_ops,
                                                       Launch L) {
    const size_t gid = blockIdx.x * (size_t)64 + threadIdx.x;
    if (gid >= (size_t)n_ops * L.n) return;
    do_rowop(Vals{v, err, L.stride, L.instance(gid)}, ops[gid / L.n], t, row_slots);
}

// PoseidonGate generator, 16 lanes per row and instance: lane l < 12 owns state element l (the latency form of the prover's tree
// kernels: a row is ~13 us of dependent instructions instead of ~65 us with one lane per row -- the step circuit's witness is a chain
// of 2 049 such rows).  Every lane of the group runs the shuffles; lanes 12..15 carry zeros.
__device__ __forceinline__ u64 shfl64(u64 x, unsigned src_lane) {
    const u32 lo = (u32)__shfl((int)(u32)x, (int)src_lane, 64), hi = (u32)__shfl((int)(u32)(x >> 32), (int)src_lane, 64);
    return ((u64)hi << 32) | lo;
}

template <class V> __device__ void poseidon_generate_wide(const V& a, const u32* rs) {
    const unsigned lane = threadIdx.x & 63u, l = lane & 15u, base = lane & ~15u;
    const bool own = l < 12;
    const unsigned lm = own ? l : 0;
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const u64 swap = a.get(rs[24]);
    if (swap > 1) {  // the same for every lane of the group
        if (l == 0) atomicOr(a.err, DE_NOT_BOOLEAN);
        return;
    }
    u64 s = own ? a.get(rs[l]) : 0;
    const u64 rhs = shfl64(s, base + ((l + 4) & 15u));
    const u64 delta = l < 4 ? gl::mul(swap, gl::sub(rhs, s)) : 0;   // swap * (rhs - lhs): lanes 0..3
    const u64 delta_lo = shfl64(delta, base + ((l + 12) & 15u));     // lanes 4..7 see the delta of lane l - 4
    if (l < 4) {
        a.set(rs[25 + l], delta);
        s = gl::add(s, delta);
    } else if (l < 8) {
        s = gl::sub(s, delta_lo);
    }
    if (own) s = gl::add_nc(s, poseidon::rc((int)l));
    for (int round = 0; round < 30; ++round) {
        const bool full = round < 4 || round >= 26;
        if (own) {
            if (round >= 1 && round < 4) a.set(rs[29 + 12 * (round - 1) + l], gl::canon(s));
            else if (round >= 26) a.set(rs[87 + 12 * (round - 26) + l], gl::canon(s));
            else if (!full && l == 0) a.set(rs[65 + (round - 4)], gl::canon(s));
        }
        if (full || l == 0) s = poseidon::sbox(s);
        u64 acc_lo = 0, acc_hi = 0;  // row lm of the MDS matrix in 32-bit halves
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            unsigned src = lm + i;
            if (src >= 12) src -= 12;
            const u64 x = shfl64(s, base + src);
            acc_lo += (u64)(u32)x * C[i];
            acc_hi += (x >> 32) * C[i];
        }
        if (l == 0) {  // MDS_MATRIX_DIAG[0] = 8
            acc_lo += (u64)(u32)s * 8;
            acc_hi += (s >> 32) * 8;
        }
        const u64 k = round + 1 < 30 ? poseidon::rc(12 * (round + 1) + (int)lm) : 0;
        acc_lo += (u32)k;
        acc_hi += k >> 32;
        s = own ? poseidon::fold96(acc_lo, acc_hi) : 0;
    }
    if (own) a.set(rs[12 + l], gl::canon(s));
}

__global__ void __launch_bounds__(64) wd_poseidon_kernel(u64* v, unsigned* err, const u32* rows, const u32* row_slots, u32 n_ops, Launch L) {
    const size_t group = (blockIdx.x * (size_t)64 + threadIdx.x) >> 4;
    if (group >= (size_t)n_ops * L.n) return;
    poseidon_generate_wide(Vals{v, err, L.stride, L.instance(group)}, row_slots + rows[group / L.n]);
}

// The tail of the schedule where every level holds PoseidonGate rows only (the hash chain): instances are independent of each other,
// so one group per instance walks the levels by itself -- one launch instead of one per level.
__global__ void __launch_bounds__(64) wd_poseidon_chain_kernel(u64* v, unsigned* err, const u32* rows, const u32* level_off, u32 first_level,
                                                                u32 last_level, const u32* row_slots, Launch L) {
    const size_t group = (blockIdx.x * (size_t)64 + threadIdx.x) >> 4;
    if (group >= L.n) return;
    const Vals a{v, err, L.stride, L.instance(group)};
    for (u32 level = first_level; level <= last_level; ++level) {
        for (u32 op = level_off[level]; op < level_off[level + 1]; ++op) poseidon_generate_wide(a, row_slots + rows[op]);
        __threadfence_block();  // the next level reads what this one stored
    }
}

template <class V> __device__ __forceinline__ void do_misc(const V& a, const Plan::MiscOp& op, const u32* aux) {
    unsigned* err = a.err;
    const u32 *in = aux + op.at, *out = in + op.n_in;
    switch (op.kind) {
        case VPBS_GEN_EQUALITY: {
            const u64 x = a.get(in[0]), y = a.get(in[1]);
            a.set(out[0], x == y ? 1 : 0);
            a.set(out[1], x == y ? 0 : gl::inv(gl::sub(x, y)));
            break;
        }
        case VPBS_GEN_BASE_SUM: {
            u64 sum = 0;
            for (u32 k = op.n_in; k-- > 0;) sum = gl::add(gl::mul(sum, op.p0), a.get(in[k]));
            a.set(out[0], sum);
            break;
        }
        case VPBS_GEN_QUOTIENT_EXT: {
            const A num{a.get(in[0]), a.get(in[1])}, den{a.get(in[2]), a.get(in[3])};
            if (den.a == 0 && den.b == 0) {
                atomicOr(err, DE_DIV_ZERO);
                break;
            }
            const u64 norm = gl::sub(gl::mul(den.a, den.a), gl::mul(7, gl::mul(den.b, den.b)));
            const A q = gates::scale(num * A{den.a, gl::neg(den.b)}, gl::inv(norm));
            a.set(out[0], q.a);
            a.set(out[1], q.b);
            break;
        }
        case VPBS_GEN_COPY: a.set(out[0], a.get(in[0])); break;
        case VPBS_GEN_LOW_HIGH: {
            const u64 x = a.get(in[0]);
            a.set(out[0], x & (((u64)1 << op.p0) - 1));
            a.set(out[1], x >> op.p0);
            break;
        }
        case 0xC0u: a.set(out[0], (u64)out[1] | ((u64)out[2] << 32)); break;  // a ConstantGate wire that is also set elsewhere
        default: break;
    }
}
__global__ void __launch_bounds__(WT) wd_misc_kernel(u64* v, unsigned* err, const Plan::MiscOp* ops, const u32* aux, u32 n_ops, Launch L) {
    const size_t gid = blockIdx.x * (size_t)WT + threadIdx.x;
    if (gid >= (size_t)n_ops * L.n) return;
    do_misc(Vals{v, err, L.stride, L.instance(gid)}, ops[gid / L.n], aux);
}

// A whole schedule walked by a FEW workgroups, a barrier between levels: for the late phase of a single instance (~160 levels of a few
// hundred operations each) the per-level launches of the batch form cost 6.9 ms of launch latency; here a level costs its slowest
// operation.  Round 4 walked it with ONE workgroup of 512 threads: 9.9 ms alone on the device (profiles/r06_few_cpus.json) -- 228 passes of 32
// PoseidonGate rows in the 16-lane form (13 us each) and, on every level, ~55 row operations of ONE thread each whose operands came through
// volatile loads one round trip at a time.  Now WALK_GROUPS workgroups share a level (256 PoseidonGate rows per pass: 138 passes), values are
// read with ordinary loads, and the barrier is the walk's own: a counter in device memory that only grows (group g adds one per level and
// waits for groups x level), release before / acquire after at device scope.  The groups need not start together -- an early one spins at its
// first barrier until the last is placed (the other streams' kernels do not wait for the walk, so their workgroups retire and make room).
constexpr unsigned WALK_THREADS = 512, WALK_GROUPS = 8;
struct WalkOffsets {
    const u32 *arith, *bits, *poseidon, *misc, *rowops;   // [n_levels + 2] each
};
__device__ __forceinline__ void walk_barrier(unsigned* counter, unsigned groups, unsigned& round) {
    if (groups == 1) {    // one workgroup = one CU: its L1 serves every wave of it, the stores only have to be done before the barrier releases
        __threadfence_block();
        __syncthreads();
        return;
    }
    __threadfence();      // this thread's stores are visible device-wide before its group arrives
    __syncthreads();
    ++round;
    if (groups > 1) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < groups * round) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        __threadfence();  // acquire for every thread of the group: ordinary loads behind the barrier must not hit lines cached before it
    }
}
__global__ void __launch_bounds__(WALK_THREADS) wd_walk_kernel(u64* v, unsigned* err, const Plan::ConstOp* consts, u32 n_consts, const Plan::ArithOp* arith,
                                                                const Plan::BitsOp* bits, const u32* poseidon, const Plan::MiscOp* misc,
                                                                const Plan::RowOp* rowops, const u32* aux, const u32* row_slots, WalkOffsets off,
                                                                RowTables t, u32 n_levels, Launch L, unsigned* counter) {
    // the walk is latency: a few waves whose every level waits for its slowest lane, on CUs they share with the other chains' hash kernels
    // (four waves per SIMD that only want throughput).  Highest wave priority: the CU's arbiter issues the walk's instructions first.
    __builtin_amdgcn_s_setprio(3);
    const unsigned groups = gridDim.x, threads = groups * WALK_THREADS, tid = blockIdx.x * WALK_THREADS + threadIdx.x;
    unsigned round = 0;
    const ValsT<true> a{v, err, L.stride, L.instance(0)};   // L.n == 1
    for (u32 i = tid; i < n_consts; i += threads) a.set(consts[i].out, consts[i].value);
    walk_barrier(counter, groups, round);
    for (u32 l = 1; l <= n_levels; ++l) {
        // the PoseidonGate rows first in the thread numbering (16 lanes each, whole groups of 16 from thread 0 on), the one-thread operations
        // from the LAST thread downwards: on a level that has both, the long single-thread generators do not queue behind a row in their wave
        for (u32 i = off.poseidon[l] + (tid >> 4); i < off.poseidon[l + 1]; i += threads / 16) poseidon_generate_wide(a, row_slots + poseidon[i]);
        const unsigned rt = threads - 1 - tid;
        for (u32 i = off.rowops[l] + rt; i < off.rowops[l + 1]; i += threads) do_rowop(a, rowops[i], t, row_slots);
        for (u32 i = off.arith[l] + rt; i < off.arith[l + 1]; i += threads) do_arith(a, arith[i]);
        for (u32 i = off.bits[l] + rt; i < off.bits[l + 1]; i += threads) do_bits(a, bits[i], aux);
        for (u32 i = off.misc[l] + rt; i < off.misc[l + 1]; i += threads) do_misc(a, misc[i], aux);
        walk_barrier(counter, groups, round);
    }
}

// one instance's column of the batch's value array as an array of its own (and the slots only the late phase writes, back): the batch layout
// val[slot][instance] makes every access of a single-instance walk a line of its own (stride = the batch); the staged late phase therefore
// works on a compact copy -- a PoseidonGate row's 110 private wires are then 880 consecutive bytes -- and the wires are gathered from it
__global__ void __launch_bounds__(WT) wd_column_kernel(const u64* __restrict__ v, u32 batch, u32 b, size_t n, u64* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)WT + threadIdx.x;
    if (i < n) out[i] = v[i * batch + b];
}
__global__ void __launch_bounds__(WT) wd_column_back_kernel(u64* __restrict__ v, u32 batch, u32 b, const u32* __restrict__ slots, size_t n, const u64* __restrict__ one) {
    const size_t i = blockIdx.x * (size_t)WT + threadIdx.x;
    if (i < n) v[(size_t)slots[i] * batch + b] = one[slots[i]];
}

// wires[pos] = val[slot][b] for every position that carries a slot (the matrix is zeroed first)
__global__ void __launch_bounds__(WT) wd_gather_kernel(const u64* v, const u32* pos, const u32* slot, size_t count, u32 batch, u32 b, u64* wires) {
    const size_t i = blockIdx.x * (size_t)WT + threadIdx.x;
    if (i < count) wires[pos[i]] = v[(size_t)slot[i] * batch + b];
}

__global__ void __launch_bounds__(WT) wd_read_kernel(const u64* v, const u32* slots, u32 count, u32 batch, u32 b, u64* out) {
    const u32 i = blockIdx.x * WT + threadIdx.x;
    if (i < count) out[i] = v[(size_t)slots[i] * batch + b];
}

template <class T> T* upload(vpbs_ctx* c, const std::vector<T>& h, std::vector<void*>& owned) {
    if (h.empty()) return nullptr;
    void* d = c->alloc_bytes(h.size() * sizeof(T));
    owned.push_back(d);
    VPBS_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return static_cast<T*>(d);
}
}  // namespace
}  // namespace vpbs

namespace vpbs {
namespace {
// the device copies of one DeviceSchedule's arrays
struct DevSched {
    const Plan::ArithOp* arith = nullptr;
    const Plan::ConstOp* consts = nullptr;
    const Plan::BitsOp* bits = nullptr;
    const Plan::MiscOp* misc = nullptr;
    const u32 *poseidon = nullptr, *aux = nullptr, *row_slots = nullptr, *preset_slot = nullptr, *poseidon_off = nullptr;
    const u32 *arith_off = nullptr, *bits_off = nullptr, *misc_off = nullptr, *rowops_off = nullptr;   // for the one-workgroup walk
    const Plan::RowOp* rowops = nullptr;
    unsigned tail_first = 0;        // levels >= tail_first hold PoseidonGate rows only (0: no such tail)
    bool preset_compares = false;   // some class is preset twice: the second preset compares, in a launch after the writers
};
}  // namespace
}  // namespace vpbs

struct vpbs_witness_device {
    vpbs_ctx* ctx = nullptr;
    const vpbs_witness_plan* plan = nullptr;
    const vpbs_witness_plan::DeviceSchedule* ds = nullptr;   // plan->dev, or plan->dev_early (the early phase of a split plan alone)
    vpbs::DevSched k;                                        // ... on the device
    const vpbs::u32* late_in = nullptr;                      // early-only objects: the slots the host's late phase wants back (device copy)
    // early-only objects: the LATE phase as a schedule of its own, run for one instance of the batch at a time (vpbs_witness_device_run_late)
    bool has_late = false;
    vpbs::DevSched k_late;
    vpbs::u32* d_instance = nullptr;   // which instance the late graph works on (read by its kernels)
    // the late phase stage by stage (plan->dev_late_stage): one schedule per stage over a shared row-slot table, the stage's presets as runs of
    // preset indices (only those words are uploaded), the values buffer the preset kernels read, and per instance how many stages are queued
    std::vector<vpbs::DevSched> k_stage;
    std::vector<std::vector<std::pair<vpbs::u32, vpbs::u32>>> stage_runs;   // [stage][(first preset index, count)]
    std::vector<unsigned> stage_groups;                                     // workgroups of the stage's walk
    vpbs::u64* d_stage_vals = nullptr;                                      // [n_preset]
    vpbs::u64* h_stage_vals = nullptr;                                      // [n_preset] pinned: a stage's words pass through it, so that queuing
                                                                            // a stage never waits for the stream (a copy from pageable memory may)
    std::vector<unsigned> stages_queued;                                    // [max_batch]
    vpbs::u64* val_one = nullptr;                                           // [n_slots + 1]: the compact copy the staged late phase works on
    int one_instance = -1;                                                  // whose column it holds (-1: nobody's)
    const vpbs::u32* d_late_slots = nullptr;                                // the slots only the late phase writes (copied back after the last stage)
    size_t n_late_slots = 0;
    hipGraphExec_t late_graph = nullptr;
    unsigned late_graph_stride = 0;
    unsigned max_batch = 0, batch = 0;
    std::vector<void*> owned;
    vpbs::u64* val = nullptr;
    unsigned* err = nullptr;
    const vpbs::u32 *out_pos = nullptr, *out_slot = nullptr;
    vpbs::RowTables tables{};
    hipGraphExec_t graph = nullptr;   // the level launches of one run for `graph_batch` instances
    unsigned graph_batch = 0;
    std::mutex mu;                    // run / wires / read share the context's stream and memory pool: one at a time per object
};

namespace vpbs {
namespace {
void launch_levels(vpbs_witness_device* d, const Plan::DeviceSchedule& ds, const DevSched& k, hipStream_t s, Launch L) {
    auto blocks = [&](size_t ops, unsigned threads) { return dim3((unsigned)((ops * L.n + threads - 1) / threads)); };
    if (!ds.consts.empty())
        hipLaunchKernelGGL(wd_const_kernel, blocks(ds.consts.size(), WT), dim3(WT), 0, s, d->val, d->err, k.consts, (u32)ds.consts.size(), L);
    const u32 last_stepwise = k.tail_first ? k.tail_first - 1 : ds.n_levels;
    for (u32 l = 1; l <= last_stepwise; ++l) {
        if (const u32 c = ds.arith_off[l + 1] - ds.arith_off[l])
            hipLaunchKernelGGL(wd_arith_kernel, blocks(c, WT), dim3(WT), 0, s, d->val, d->err, k.arith + ds.arith_off[l], c, L);
        if (const u32 c = ds.bits_off[l + 1] - ds.bits_off[l])
            hipLaunchKernelGGL(wd_bits_kernel, blocks(c, WT), dim3(WT), 0, s, d->val, d->err, k.bits + ds.bits_off[l], k.aux, c, L);
        if (const u32 c = ds.poseidon_off[l + 1] - ds.poseidon_off[l])
            hipLaunchKernelGGL(wd_poseidon_kernel, blocks((size_t)c * 16, 64), dim3(64), 0, s, d->val, d->err, k.poseidon + ds.poseidon_off[l],
                               k.row_slots, c, L);
        if (const u32 c = ds.misc_off[l + 1] - ds.misc_off[l])
            hipLaunchKernelGGL(wd_misc_kernel, blocks(c, WT), dim3(WT), 0, s, d->val, d->err, k.misc + ds.misc_off[l], k.aux, c, L);
        if (const u32 c = ds.rowops_off[l + 1] - ds.rowops_off[l])
            hipLaunchKernelGGL(wd_rowop_kernel, blocks(c, 64), dim3(64), 0, s, d->val, d->err, k.rowops + ds.rowops_off[l], d->tables, k.row_slots, c, L);
    }
    if (k.tail_first)
        hipLaunchKernelGGL(wd_poseidon_chain_kernel, dim3((L.n * 16u + 63) / 64), dim3(64), 0, s, d->val, d->err, k.poseidon, k.poseidon_off,
                           k.tail_first, ds.n_levels, k.row_slots, L);
}
}  // namespace
}  // namespace vpbs

extern "C" {

extern "C++" {
namespace vpbs {
namespace {
DevSched upload_schedule(vpbs_ctx* ctx, const Plan::DeviceSchedule& ds, std::vector<void*>& owned) {
    DevSched k;
    k.arith = upload(ctx, ds.arith, owned);
    k.consts = upload(ctx, ds.consts, owned);
    k.bits = upload(ctx, ds.bits, owned);
    k.misc = upload(ctx, ds.misc, owned);
    k.poseidon = upload(ctx, ds.poseidon, owned);
    k.aux = upload(ctx, ds.aux, owned);
    k.row_slots = upload(ctx, ds.row_slots, owned);
    k.preset_slot = upload(ctx, ds.preset_slot, owned);
    for (u32 sl : ds.preset_slot) k.preset_compares |= (sl & Plan::CHECK) != 0 && (sl & ~Plan::CHECK) != 0xFFFFFFFFu;
    k.poseidon_off = upload(ctx, ds.poseidon_off, owned);
    k.arith_off = upload(ctx, ds.arith_off, owned);
    k.bits_off = upload(ctx, ds.bits_off, owned);
    k.misc_off = upload(ctx, ds.misc_off, owned);
    k.rowops_off = upload(ctx, ds.rowops_off, owned);
    k.rowops = upload(ctx, ds.rowops, owned);
    u32 l = ds.n_levels;
    while (l >= 1 && ds.arith_off[l + 1] == ds.arith_off[l] && ds.bits_off[l + 1] == ds.bits_off[l] && ds.misc_off[l + 1] == ds.misc_off[l] &&
           ds.rowops_off[l + 1] == ds.rowops_off[l])
        --l;
    k.tail_first = ds.n_levels - l >= 8 ? l + 1 : 0;
    return k;
}
}  // namespace
}  // namespace vpbs
}  // extern "C++"

static int device_create(vpbs_ctx* ctx, const vpbs_witness_plan* plan, unsigned max_batch, bool early, vpbs_witness_device** out) {
    if (!ctx || !plan || !out || max_batch == 0 || (early && !plan->is_split)) return VPBS_ERR_INVALID;
    try {
        VPBS_HIP(hipSetDevice(ctx->device));
        const auto& ds = early ? plan->dev_early : plan->dev;
        VPBS_REQUIRE(ds.supported, ("this circuit has no device witness generator: " + ds.unsupported).c_str());
        auto d = std::make_unique<vpbs_witness_device>();
        d->ctx = ctx;
        d->plan = plan;
        d->ds = &ds;
        d->max_batch = max_batch;
        using namespace vpbs;
        d->k = upload_schedule(ctx, ds, d->owned);
        if (early) {
            d->late_in = upload(ctx, plan->late_in_slots, d->owned);
            if (plan->dev_late.supported) {
                d->k_late = upload_schedule(ctx, plan->dev_late, d->owned);
                d->d_instance = static_cast<u32*>(ctx->alloc_bytes(sizeof(u32)));
                d->owned.push_back(d->d_instance);
                d->has_late = true;
                if (!plan->dev_late_stage.empty()) {
                    const u32* shared = upload(ctx, plan->late_row_slots, d->owned);
                    for (size_t k = 0; k < plan->dev_late_stage.size(); ++k) {
                        const auto& sd = plan->dev_late_stage[k];
                        DevSched ks = upload_schedule(ctx, sd, d->owned);
                        ks.row_slots = shared;
                        d->k_stage.push_back(ks);
                        std::vector<std::pair<u32, u32>> runs;
                        for (u32 i : plan->stage_presets[k]) {
                            if (!runs.empty() && runs.back().first + runs.back().second == i) ++runs.back().second;
                            else runs.push_back({i, 1u});
                        }
                        d->stage_runs.push_back(std::move(runs));
                        // a thin stage (the transcript's chain: a handful of operations on each of its levels) is walked by ONE workgroup --
                        // its levels then cost a workgroup barrier, not the device-wide one; a stage with wide levels by all of them
                        // (by the AVERAGE lanes a level wants: stage 1 of the cyclic circuit has one level of a thousand operations among 162
                        // that want fifty -- measured alone on the device, eight workgroups 3.37 ms, one workgroup with block-scope fences less)
                        const size_t lanes = 16 * sd.poseidon.size() + sd.arith.size() + sd.rowops.size() + sd.misc.size() + sd.bits.size();
                        d->stage_groups.push_back(sd.n_levels && lanes / sd.n_levels > 2 * WALK_THREADS ? WALK_GROUPS : 1u);
                    }
                    d->d_stage_vals = ctx->alloc_words(std::max<size_t>(1, plan->preset_slot.size()));
                    d->owned.push_back(d->d_stage_vals);
                    VPBS_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->h_stage_vals), sizeof(u64) * (plan->preset_slot.size() + 1), hipHostMallocDefault));   // + one word: the instance number on its way to the device
                    d->stages_queued.assign(max_batch, 0);
                    d->val_one = ctx->alloc_words(plan->n_slots + 1);
                    d->owned.push_back(d->val_one);
                    std::vector<u32> late_slots;
                    for (const auto& run : plan->late_slot_runs)
                        for (u32 i = 0; i < run.second; ++i) late_slots.push_back(run.first + i);
                    d->n_late_slots = late_slots.size();
                    d->d_late_slots = upload(ctx, late_slots, d->owned);
                }
            }
        }
        d->out_pos = upload(ctx, plan->out_pos, d->owned);
        d->out_slot = upload(ctx, plan->out_slot, d->owned);
        if (!ds.rowops.empty() || (d->has_late && !plan->dev_late.rowops.empty())) {
            std::vector<gates::CosetTables> coset(plan->gates.size());
            for (size_t i = 0; i < plan->gates.size(); ++i)
                if (plan->gates[i].kind == VPBS_GATE_COSET_INTERPOLATION) coset[i] = gates::coset_tables(plan->gates[i].p0);
            d->tables = RowTables{upload(ctx, plan->gates, d->owned), upload(ctx, plan->row_gate, d->owned), upload(ctx, plan->row_off, d->owned),
                                  upload(ctx, plan->consts, d->owned), upload(ctx, coset, d->owned), std::max(1u, plan->max_consts)};
        }
        d->val = ctx->alloc_words((plan->n_slots + 1) * (size_t)max_batch);   // + the scratch slot the other phase's presets are routed to
        d->owned.push_back(d->val);
        d->err = static_cast<unsigned*>(ctx->alloc_bytes(4 * sizeof(unsigned)));   // flags, first conflicting slot + 1, the walk's barrier counter, spare
        d->owned.push_back(d->err);
        VPBS_HIP(vpbs::stream_sync(ctx->stream));
        *out = d.release();
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        ctx->err = e.what;
        return e.status;
    }
}

int vpbs_witness_device_create(vpbs_ctx* ctx, const vpbs_witness_plan* plan, unsigned max_batch, vpbs_witness_device** out) {
    return device_create(ctx, plan, max_batch, false, out);
}
int vpbs_witness_device_create_early(vpbs_ctx* ctx, const vpbs_witness_plan* plan, unsigned max_batch, vpbs_witness_device** out) {
    return device_create(ctx, plan, max_batch, true, out);
}

void vpbs_witness_device_free(vpbs_witness_device* d) {
    if (!d) return;
    (void)hipSetDevice(d->ctx->device);
    (void)vpbs::stream_sync(d->ctx->stream);
    if (d->graph) (void)hipGraphExecDestroy(d->graph);
    if (d->late_graph) (void)hipGraphExecDestroy(d->late_graph);
    if (d->h_stage_vals) (void)hipHostFree(d->h_stage_vals);
    for (void* p : d->owned) d->ctx->release(p);
    delete d;
}

extern "C++" {
namespace vpbs {
namespace {
void throw_on_flags(const vpbs_witness_device* d, const unsigned report[2]) {
    const unsigned flags = report[0];
    if (!flags) return;
    std::string m;
    if (flags & DE_SET_TWICE) {
        m += "a partition was set twice with different values";
        if (report[1]) {   // a wire of that class: the first position that carries the slot
            const u32 slot = report[1] - 1;
            for (size_t i = 0; i < d->plan->out_slot.size(); ++i)
                if (d->plan->out_slot[i] == slot) {
                    m += " (the class of wire column " + std::to_string(d->plan->out_pos[i] / d->plan->n) + ", row " +
                         std::to_string(d->plan->out_pos[i] % d->plan->n) + ")";
                    break;
                }
        }
        m += "; ";
    }
    if (flags & DE_TOO_LARGE) m += "an integer too large to fit in the given number of limbs; ";
    if (flags & DE_NOT_BOOLEAN) m += "PoseidonGate: swap wire is not boolean; ";
    if (flags & DE_DIV_ZERO) m += "QuotientGeneratorExtension: division by zero; ";
    if (flags & DE_GATE) m += "a gate generator rejected its inputs (limbs that do not fit, an access index out of range, a non-boolean bit, a zero shift); ";
    throw DeviceError{VPBS_ERR_INVALID, "device witness generation: " + m.substr(0, m.size() - 2)};
}
// presets -> (captured) level launches -> error flags, on the context's stream; returns after the stream has drained
void run_schedule(vpbs_witness_device* d, const Plan::DeviceSchedule& ds, const DevSched& k, hipGraphExec_t& graph, unsigned& graph_key, unsigned key,
                  Launch L, const u64* preset_val, u64*& d_vals, bool walk = false) {
    vpbs_ctx* ctx = d->ctx;
    hipStream_t s = ctx->stream;
    const size_t n_preset = d->plan->preset_slot.size();
    VPBS_HIP(hipMemsetAsync(d->err, 0, 4 * sizeof(unsigned), s));   // flags, first conflicting slot + 1, the walk's barrier counter, spare
    if (n_preset) {
        d_vals = ctx->alloc_words(n_preset * L.n);
        VPBS_HIP(hipMemcpyAsync(d_vals, preset_val, sizeof(u64) * n_preset * L.n, hipMemcpyHostToDevice, s));
        for (u32 pass = 0; pass < (k.preset_compares ? 2u : 1u); ++pass)
            hipLaunchKernelGGL(wd_preset_kernel, dim3((unsigned)((n_preset * L.n + WT - 1) / WT)), dim3(WT), 0, s, d->val, d->err, k.preset_slot, d_vals,
                               (u32)n_preset, L, pass);
    }
    if (walk) {   // one instance: a few workgroups walk the levels (VPBS_WALK_GROUPS: 1 .. 32, default 8; a development switch)
        static const unsigned groups = [] {
            const char* e = std::getenv("VPBS_WALK_GROUPS");
            const long g = e ? std::strtol(e, nullptr, 10) : (long)WALK_GROUPS;
            return (unsigned)std::min(32l, std::max(1l, g));
        }();
        hipLaunchKernelGGL(wd_walk_kernel, dim3(groups), dim3(WALK_THREADS), 0, s, d->val, d->err, k.consts, (u32)ds.consts.size(), k.arith, k.bits, k.poseidon,
                           k.misc, k.rowops, k.aux, k.row_slots, WalkOffsets{k.arith_off, k.bits_off, k.poseidon_off, k.misc_off, k.rowops_off}, d->tables,
                           ds.n_levels, L, d->err + 2);
        VPBS_HIP(hipGetLastError());
    } else if (!graph || graph_key != key) {   // the level launches are a static sequence: captured once per batch size (stride), replayed afterwards
        if (graph) {
            VPBS_HIP(hipGraphExecDestroy(graph));
            graph = nullptr;
        }
        hipGraph_t g = nullptr;
        VPBS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        launch_levels(d, ds, k, s, L);
        const hipError_t launched = hipGetLastError();
        const hipError_t ended = hipStreamEndCapture(s, &g);   // always leave capture mode
        hipError_t e = launched != hipSuccess ? launched : ended;
        if (e == hipSuccess) e = hipGraphInstantiate(&graph, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (e != hipSuccess) graph = nullptr;
        VPBS_HIP(e);
        graph_key = key;
    }
    if (!walk) VPBS_HIP(hipGraphLaunch(graph, s));
    unsigned report[2] = {0, 0};
    VPBS_HIP(hipMemcpyAsync(report, d->err, sizeof report, hipMemcpyDeviceToHost, s));
    VPBS_HIP(vpbs::stream_sync(s));
    if (d_vals) ctx->release(d_vals);
    d_vals = nullptr;
    throw_on_flags(d, report);
}
}  // namespace
}  // namespace vpbs
}  // extern "C++"

int vpbs_witness_device_run(vpbs_witness_device* d, const uint64_t* preset_val, unsigned batch) {
    if (!d || batch == 0 || batch > d->max_batch || (!d->plan->preset_slot.empty() && !preset_val)) return VPBS_ERR_INVALID;
    vpbs_ctx* ctx = d->ctx;
    std::lock_guard<std::mutex> lock(d->mu);
    vpbs::u64* d_vals = nullptr;
    try {
        using namespace vpbs;
        VPBS_HIP(hipSetDevice(ctx->device));
        d->batch = batch;
        d->one_instance = -1;
        std::fill(d->stages_queued.begin(), d->stages_queued.end(), 0u);   // a new batch: no instance has late stages in flight (a chain that gave up may have left some)
        VPBS_HIP(hipMemsetAsync(d->val, 0, sizeof(u64) * (d->plan->n_slots + 1) * batch, ctx->stream));
        run_schedule(d, *d->ds, d->k, d->graph, d->graph_batch, batch, Launch{batch, batch, nullptr}, preset_val, d_vals);
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        if (d_vals) {
            (void)vpbs::stream_sync(ctx->stream);
            ctx->release(d_vals);
        }
        ctx->err = e.what;
        return e.status;
    }
}

int vpbs_witness_device_has_late(const vpbs_witness_device* d) { return d && d->has_late ? 1 : 0; }

}  // extern "C"
// The late phase of one instance STAGE BY STAGE (library-internal: the IVC driver's form; the C ABI's vpbs_witness_device_run_late runs the
// stages back to back).  Stage numbers as in vpbs_witness_plan_split; a stage may be queued as soon as its presets -- that section of the
// previous proof -- are final in preset_val: only those words are uploaded, the stage's presets are put in place and its levels walked, all on
// the object's stream, and with wait = 0 the call returns at once (~20 us of enqueues).  Stages must be queued in order; the call for the LAST
// stage (or any call with wait != 0) waits for everything queued and reports what the generators found.
namespace vpbs {
unsigned witness_device_late_stages(const vpbs_witness_device* d) { return d && d->has_late ? (unsigned)d->k_stage.size() : 0; }
int witness_device_run_late_stage(vpbs_witness_device* d, unsigned instance, unsigned stage, const uint64_t* preset_val, int wait) {
    if (!d || !d->has_late || d->k_stage.empty() || !preset_val || stage < 1 || stage > d->k_stage.size()) return VPBS_ERR_INVALID;
    vpbs_ctx* ctx = d->ctx;
    std::lock_guard<std::mutex> lock(d->mu);
    if (instance >= d->batch || d->stages_queued[instance] != stage - 1) return VPBS_ERR_INVALID;
    try {
        VPBS_HIP(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        const DevSched& k = d->k_stage[stage - 1];
        const Plan::DeviceSchedule& ds = d->plan->dev_late_stage[stage - 1];
        const Launch L{1, 1, nullptr};   // the compact copy: stride 1, instance 0
        const size_t n_val = d->plan->n_slots + 1;
        if (stage == 1) {
            hipLaunchKernelGGL(wd_column_kernel, dim3((unsigned)((n_val + WT - 1) / WT)), dim3(WT), 0, s, d->val, d->batch, instance, n_val, d->val_one);
            d->one_instance = (int)instance;
            VPBS_HIP(hipMemsetAsync(d->err, 0, 2 * sizeof(unsigned), s));   // the flags of the whole late phase
        }
        if (d->one_instance != (int)instance) throw DeviceError{VPBS_ERR_INVALID, "late stages of two instances interleaved on one device witness object"};
        VPBS_HIP(hipMemsetAsync(d->err + 2, 0, 2 * sizeof(unsigned), s));   // the walk's barrier counter
        for (const auto& run : d->stage_runs[stage - 1]) {
            std::memcpy(d->h_stage_vals + run.first, preset_val + run.first, sizeof(u64) * run.second);
            VPBS_HIP(hipMemcpyAsync(d->d_stage_vals + run.first, d->h_stage_vals + run.first, sizeof(u64) * run.second, hipMemcpyHostToDevice, s));
        }
        for (u32 pass = 0; pass < (k.preset_compares ? 2u : 1u); ++pass)   // every writer of the stage before the first comparer, whichever run holds it
            for (const auto& run : d->stage_runs[stage - 1])
                hipLaunchKernelGGL(wd_preset_kernel, dim3((run.second + WT - 1) / WT), dim3(WT), 0, s, d->val_one, d->err, k.preset_slot + run.first,
                                   d->d_stage_vals + run.first, run.second, L, pass);
        if (ds.n_levels || !ds.consts.empty())
            hipLaunchKernelGGL(wd_walk_kernel, dim3(d->stage_groups[stage - 1]), dim3(WALK_THREADS), 0, s, d->val_one, d->err, k.consts, (u32)ds.consts.size(),
                               k.arith, k.bits, k.poseidon, k.misc, k.rowops, k.aux, k.row_slots,
                               WalkOffsets{k.arith_off, k.bits_off, k.poseidon_off, k.misc_off, k.rowops_off}, d->tables, ds.n_levels, L, d->err + 2);
        VPBS_HIP(hipGetLastError());
        const bool last = stage == d->k_stage.size();
        d->stages_queued[instance] = last ? 0 : stage;
        if (last && d->n_late_slots)   // the batch's array holds the instance's final values as well (readers of single positions, a later gather)
            hipLaunchKernelGGL(wd_column_back_kernel, dim3((unsigned)((d->n_late_slots + WT - 1) / WT)), dim3(WT), 0, s, d->val, d->batch, instance, d->d_late_slots,
                               d->n_late_slots, d->val_one);
        if (wait || last) {
            unsigned report[2] = {0, 0};
            VPBS_HIP(hipMemcpyAsync(report, d->err, sizeof report, hipMemcpyDeviceToHost, s));
            VPBS_HIP(vpbs::stream_sync(s));
            throw_on_flags(d, report);
        }
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        d->stages_queued[instance] = 0;
        (void)vpbs::stream_sync(ctx->stream);
        ctx->err = e.what;
        return e.status;
    }
}
}  // namespace vpbs
extern "C" {

int vpbs_witness_device_run_late(vpbs_witness_device* d, unsigned instance, const uint64_t* preset_val) {
    if (!d || !d->has_late || !preset_val) return VPBS_ERR_INVALID;
    vpbs_ctx* ctx = d->ctx;
    static const bool whole = std::getenv("VPBS_DEVICE_LATE_WHOLE") != nullptr;   // A-B: the late phase as ONE schedule (round 4's form)
    if (!d->k_stage.empty() && !whole) {   // a staged plan: the stages back to back (those already queued for this instance are not repeated)
        unsigned first;
        {
            std::lock_guard<std::mutex> lock(d->mu);
            if (instance >= d->batch) return VPBS_ERR_INVALID;
            first = d->stages_queued[instance] + 1;
        }
        int rc = VPBS_OK;
        for (unsigned st = first; st <= d->k_stage.size() && rc == VPBS_OK; ++st) rc = vpbs::witness_device_run_late_stage(d, instance, st, preset_val, 0);
        return rc;
    }
    std::lock_guard<std::mutex> lock(d->mu);
    if (instance >= d->batch) return VPBS_ERR_INVALID;
    vpbs::u64* d_vals = nullptr;
    try {
        using namespace vpbs;
        VPBS_HIP(hipSetDevice(ctx->device));
        const u32 inst = instance;
        VPBS_HIP(hipMemcpyAsync(d->d_instance, &inst, sizeof inst, hipMemcpyHostToDevice, ctx->stream));   // run_schedule drains the stream before returning
        static const bool per_level = std::getenv("VPBS_DEVICE_LATE_LAUNCHES") != nullptr;   // A-B: one launch per level and kind (6.9 ms per step)
        run_schedule(d, d->plan->dev_late, d->k_late, d->late_graph, d->late_graph_stride, d->batch, Launch{d->batch, 1, d->d_instance}, preset_val,
                     d_vals, !per_level);
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        if (d_vals) {
            (void)vpbs::stream_sync(ctx->stream);
            ctx->release(d_vals);
        }
        ctx->err = e.what;
        return e.status;
    }
}

int vpbs_witness_device_wires(vpbs_witness_device* d, unsigned instance, uint64_t* d_wires) {
    if (!d || !d_wires) return VPBS_ERR_INVALID;
    vpbs_ctx* ctx = d->ctx;
    std::lock_guard<std::mutex> lock(d->mu);
    if (instance >= d->batch) return VPBS_ERR_INVALID;
    try {
        using namespace vpbs;
        VPBS_HIP(hipSetDevice(ctx->device));
        const size_t count = d->plan->out_pos.size();
        VPBS_HIP(hipMemsetAsync(d_wires, 0, sizeof(u64) * d->plan->total, ctx->stream));
        if (d->one_instance == (int)instance && d->stages_queued[instance] == 0)   // its late phase ran on the compact copy: every value is there, close together
            hipLaunchKernelGGL(wd_gather_kernel, dim3((unsigned)((count + WT - 1) / WT)), dim3(WT), 0, ctx->stream, d->val_one, d->out_pos, d->out_slot, count,
                               1u, 0u, d_wires);
        else
            hipLaunchKernelGGL(wd_gather_kernel, dim3((unsigned)((count + WT - 1) / WT)), dim3(WT), 0, ctx->stream, d->val, d->out_pos, d->out_slot, count,
                               d->batch, instance, d_wires);
        VPBS_HIP(vpbs::stream_sync(ctx->stream));
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        ctx->err = e.what;
        return e.status;
    }
}

int vpbs_witness_device_read_late_inputs(vpbs_witness_device* d, unsigned instance, uint64_t* out) {
    if (!d || !d->late_in || !out) return VPBS_ERR_INVALID;
    vpbs_ctx* ctx = d->ctx;
    std::lock_guard<std::mutex> lock(d->mu);
    if (instance >= d->batch) return VPBS_ERR_INVALID;
    const size_t count = d->plan->late_in_slots.size();
    if (count == 0) return VPBS_OK;
    vpbs::u64* d_out = nullptr;
    try {
        using namespace vpbs;
        VPBS_HIP(hipSetDevice(ctx->device));
        d_out = ctx->alloc_words(count);
        hipLaunchKernelGGL(wd_read_kernel, dim3((unsigned)((count + WT - 1) / WT)), dim3(WT), 0, ctx->stream, d->val, d->late_in, (u32)count, d->batch,
                           instance, d_out);
        VPBS_HIP(hipMemcpyAsync(out, d_out, sizeof(u64) * count, hipMemcpyDeviceToHost, ctx->stream));
        VPBS_HIP(vpbs::stream_sync(ctx->stream));
        ctx->release(d_out);
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        (void)vpbs::stream_sync(ctx->stream);
        if (d_out) ctx->release(d_out);
        ctx->err = e.what;
        return e.status;
    }
}

int vpbs_witness_device_read(vpbs_witness_device* d, unsigned instance, const uint32_t* positions, size_t count, uint64_t* out) {
    if (!d || (count && (!positions || !out))) return VPBS_ERR_INVALID;
    vpbs_ctx* ctx = d->ctx;
    std::lock_guard<std::mutex> lock(d->mu);
    if (instance >= d->batch) return VPBS_ERR_INVALID;
    vpbs::u32* d_slots = nullptr;
    vpbs::u64* d_out = nullptr;
    try {
        using namespace vpbs;
        VPBS_HIP(hipSetDevice(ctx->device));
        // positions -> slots through the plan's ascending (position, slot) list; a position without a slot reads 0
        const auto& pos = d->plan->out_pos;
        std::vector<u32> slots(count);
        std::vector<size_t> missing;
        for (size_t i = 0; i < count; ++i) {
            const auto it = std::lower_bound(pos.begin(), pos.end(), positions[i]);
            if (it != pos.end() && *it == positions[i]) slots[i] = d->plan->out_slot[it - pos.begin()];
            else {
                slots[i] = 0;
                missing.push_back(i);
            }
        }
        if (count == 0) return VPBS_OK;
        d_slots = static_cast<u32*>(ctx->alloc_bytes(sizeof(u32) * count));
        d_out = ctx->alloc_words(count);
        VPBS_HIP(hipMemcpyAsync(d_slots, slots.data(), sizeof(u32) * count, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(wd_read_kernel, dim3((unsigned)((count + WT - 1) / WT)), dim3(WT), 0, ctx->stream, d->val, d_slots, (u32)count, d->batch,
                           instance, d_out);
        VPBS_HIP(hipMemcpyAsync(out, d_out, sizeof(u64) * count, hipMemcpyDeviceToHost, ctx->stream));
        VPBS_HIP(vpbs::stream_sync(ctx->stream));
        ctx->release(d_slots);
        ctx->release(d_out);
        for (size_t i : missing) out[i] = 0;
        return VPBS_OK;
    } catch (const vpbs::DeviceError& e) {
        (void)vpbs::stream_sync(ctx->stream);   // nothing may still be using the staging blocks when they go back to the pool
        if (d_slots) ctx->release(d_slots);
        if (d_out) ctx->release(d_out);
        ctx->err = e.what;
        return e.status;
    }
}

}  // extern "C"
