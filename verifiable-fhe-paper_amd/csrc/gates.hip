// Gate-constraint terms of the quotient polynomials on gfx950, and the host-side companions (selector layout, gate
// evaluation at zeta for the verifier, per-gate witness rows).
// Replaces plonky2 0.2.0 plonk/vanishing_poly.rs `evaluate_gate_constraints_base_batch` (called from
// `eval_vanishing_poly_base_batch` inside `compute_quotient_polys`, plonk/prover.rs) -- the stage of prove() reached from
// /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364.
//
// One kernel instantiation per gate type, one thread per LDE point.  A thread reads the wires the gate uses straight from
// the committed wires LDE (column-major, leaf order => coalesced), evaluates the gate's constraints in plonky2's order,
// folds them with the powers of every alpha (uniform scalar loads), multiplies by the selector filter and accumulates
// into out[challenge][point].  The constraints of different gates share constraint indices (plonky2 adds them: at most one
// filter is non-zero on a trace row), hence the accumulation.  HBM traffic: each gate reads only its own wires once.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "context.h"
#include "gates.h"

namespace vpbs {
namespace {
constexpr unsigned THREADS = 256;
struct PiHash {
    u64 h[4];
};

struct DevVars {
    using F = u64;
    const u64* wires;
    const u64* consts;  // first gate-constant column (selectors skipped)
    size_t stride, j;
    PiHash pih;
    __device__ __forceinline__ u64 wire(unsigned i) const { return wires[(size_t)i * stride + j]; }
    __device__ __forceinline__ u64 constant(unsigned i) const { return consts[(size_t)i * stride + j]; }
    __device__ __forceinline__ u64 pi_hash(unsigned i) const { return pih.h[i]; }
};
// gl::LazyAcc (gl.h): reduce_with_powers over the constraints with the reduction mod p deferred to the end of the gate
using gl::LazyAcc;
// The sink multiplies every constraint into NC running sums WITHOUT asking how many challenges there are: with fewer than NC challenges the
// spare sums repeat the last challenge's powers (row[] points there) and are never read.  A run-time `a < nc` around each multiply-accumulate
// made the compiler keep two copies of every accumulator (three 64-bit moves per constraint and challenge) and a branch per challenge.
template <int NC> struct DevSinkT {
    const u64* row[NC];  // challenge a's powers of alpha (challenge nc - 1's for a >= nc)
    unsigned idx;
    LazyAcc acc[NC];
    __device__ __forceinline__ DevSinkT(const u64* apow, unsigned pow_stride, unsigned nc) : idx(0), acc{} {
#pragma unroll
        for (int a = 0; a < NC; ++a) row[a] = apow + (size_t)((unsigned)a < nc ? (unsigned)a : nc - 1) * pow_stride;   // nc >= 1: checked where the kernels are launched
    }
    __device__ __forceinline__ void push(u64 c) {
        const u32 c0 = (u32)c, c1 = (u32)(c >> 32);
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            const u64 p = row[a][idx];
            acc[a].mac(c0, c1, (u32)p, (u32)(p >> 32));
        }
        ++idx;
    }
};

// PoseidonMdsGate on the GPU: the MDS acts on the two components of the algebra elements separately, so it is two runs of the
// hashing kernels' multiply-add MDS layer (poseidon.h) instead of 2 x 156 modular multiplications by MDS entries.
template <class Vars, class DevSink> __device__ __forceinline__ void poseidon_mds_dev(const Vars& v, DevSink& s) {
    u64 a[12], b[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        a[i] = v.wire(2 * i);
        b[i] = v.wire(2 * i + 1);
    }
    poseidon::mds_add_const(a, nullptr);
    poseidon::mds_add_const(b, nullptr);
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        s.push(gl::sub_a(v.wire(24 + 2 * r), a[r]));
        s.push(gl::sub_a(v.wire(25 + 2 * r), b[r]));
    }
}

// PoseidonGate on the GPU (same constraints, same order as gates::eval_poseidon): the hashing kernels' round functions with the
// S-box inputs swapped for the gate's wires -- paired S-boxes, multiply-add MDS with the next round's constants folded in, and the
// 22 partial rounds as 7 fused groups of three (one dense M^3 pass per group, poseidon.h) + 1.
template <class Vars, class DevSink> __device__ __forceinline__ void poseidon_gate_dev(const Vars& v, DevSink& s) {
    using poseidon::rc;
    const u64 swap = v.wire(24);
    s.push(gl::mul(swap, gl::sub(swap, 1)));
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u64 lhs = v.wire(i), rhs = v.wire(i + 4), delta = v.wire(25 + i);
        s.push(gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta));
        st[i] = gl::add(lhs, delta);
        st[i + 4] = gl::sub(rhs, delta);
    }
#pragma unroll
    for (int i = 8; i < 12; ++i) st[i] = v.wire(i);
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = gl::add_nc(st[i], rc(i));
    for (int r = 0; r < 4; ++r) {  // first full rounds (round index r)
        u64 kc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) kc[i] = rc(12 * (r + 1) + i);
        if (r != 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const u64 in = v.wire(29 + 12 * (r - 1) + i);
                s.push(gl::sub_a(st[i], in));
                st[i] = in;
            }
        }
#pragma unroll
        for (int i = 0; i < 12; i += 2) poseidon::sbox2(st[i], st[i + 1]);
        poseidon::mds_add_const(st, kc);
    }
    for (int g = 0; g < 7; ++g) {  // partial rounds 0..20
        u64 w[3], x[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = v.wire(65 + 3 * g + i);
        s.push(gl::sub_a(st[0], w[0]));
        poseidon::partial_group3_core<true>(st, g, w, x);
        s.push(gl::sub(x[0], w[1]));
        s.push(gl::sub(x[1], w[2]));
    }
    {  // partial round 21
        u64 kc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) kc[i] = rc(12 * 26 + i);
        const u64 in = v.wire(65 + 21);
        s.push(gl::sub_a(st[0], in));
        st[0] = poseidon::sbox(in);
        poseidon::mds_add_const(st, kc);
    }
    for (int r = 0; r < 4; ++r) {  // second full rounds (round index 26 + r)
        u64 kc[12];
        const int next = r < 3 ? 12 * (27 + r) : 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) kc[i] = rc(next + i);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const u64 in = v.wire(87 + 12 * r + i);
            s.push(gl::sub_a(st[i], in));
            st[i] = in;
        }
#pragma unroll
        for (int i = 0; i < 12; i += 2) poseidon::sbox2(st[i], st[i + 1]);
        poseidon::mds_add_const(st, r < 3 ? kc : nullptr);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) s.push(gl::sub_a(v.wire(12 + i), st[i]));
}

// The PoseidonGate in independent pieces (the one-launch tile kernel gives them to different waves): every full round's S-box inputs are
// wires, so a stretch of rounds between two such wire sets needs nothing from the rounds before it.  PART 1: the S-box wires of round 3 ->
// S-box, MDS, the 22 partial rounds -> the S-box wires of round 26 (constraints 41..74); PART 2: swap / delta constraints, rounds 0..2
// against the S-box wires of rounds 1..3 (constraints 0..40); PART 3: rounds 26..29 and the outputs (constraints 75..122).  Same values, same
// constraint indices (the sink's index is set to the piece's first constraint) as poseidon_gate_dev.
template <int PART, class Vars, class DevSink> __device__ __forceinline__ void poseidon_gate_part(const Vars& v, DevSink& s) {
    using poseidon::rc;
    u64 st[12];
    if constexpr (PART == 2) {
        s.idx = 0;
        const u64 swap = v.wire(24);
        s.push(gl::mul(swap, gl::sub(swap, 1)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u64 lhs = v.wire(i), rhs = v.wire(i + 4), delta = v.wire(25 + i);
            s.push(gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta));
            st[i] = gl::add(lhs, delta);
            st[i + 4] = gl::sub(rhs, delta);
        }
#pragma unroll
        for (int i = 8; i < 12; ++i) st[i] = v.wire(i);
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = gl::add_nc(st[i], rc(i));
        for (int r = 0; r < 3; ++r) {   // round r: S-box, MDS + constants of round r + 1, compared with the S-box wires of round r + 1
            u64 kc[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) kc[i] = rc(12 * (r + 1) + i);
#pragma unroll
            for (int i = 0; i < 12; i += 2) poseidon::sbox2(st[i], st[i + 1]);
            poseidon::mds_add_const(st, kc);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const u64 in = v.wire(29 + 12 * r + i);
                s.push(gl::sub_a(st[i], in));
                st[i] = in;
            }
        }
    }
    if constexpr (PART == 1) {
        s.idx = 41;
        {
            u64 kc[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                kc[i] = rc(12 * 4 + i);
                st[i] = v.wire(29 + 24 + i);   // the S-box wires of round 3
            }
#pragma unroll
            for (int i = 0; i < 12; i += 2) poseidon::sbox2(st[i], st[i + 1]);
            poseidon::mds_add_const(st, kc);
        }
        for (int g = 0; g < 7; ++g) {  // partial rounds 0..20
            u64 w[3], x[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) w[i] = v.wire(65 + 3 * g + i);
            s.push(gl::sub_a(st[0], w[0]));
            poseidon::partial_group3_core<true>(st, g, w, x);
            s.push(gl::sub(x[0], w[1]));
            s.push(gl::sub(x[1], w[2]));
        }
        {  // partial round 21
            u64 kc[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) kc[i] = rc(12 * 26 + i);
            const u64 in = v.wire(65 + 21);
            s.push(gl::sub_a(st[0], in));
            st[0] = poseidon::sbox(in);
            poseidon::mds_add_const(st, kc);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) s.push(gl::sub_a(st[i], v.wire(87 + i)));
    }
    if constexpr (PART == 3) {
        s.idx = 75;
        for (int r = 0; r < 4; ++r) {  // round 26 + r from its S-box wires; compared with the next round's wires, the last with the outputs
            u64 kc[12];
            const int next = r < 3 ? 12 * (27 + r) : 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                kc[i] = rc(next + i);
                st[i] = v.wire(87 + 12 * r + i);
            }
#pragma unroll
            for (int i = 0; i < 12; i += 2) poseidon::sbox2(st[i], st[i + 1]);
            poseidon::mds_add_const(st, r < 3 ? kc : nullptr);
            if (r < 3) {
#pragma unroll
                for (int i = 0; i < 12; ++i) s.push(gl::sub_a(st[i], v.wire(87 + 12 * (r + 1) + i)));
            }
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) s.push(gl::sub_a(v.wire(12 + i), st[i]));
    }
}

template <unsigned KIND, class Vars, class DevSink>
__device__ __forceinline__ void eval_kind(const vpbs_gate& g, const gates::CosetTables& t, const Vars& v, DevSink& s) {
    if constexpr (KIND == VPBS_GATE_CONSTANT) gates::eval_constant<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_PUBLIC_INPUT) gates::eval_public_input<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_ARITHMETIC) gates::eval_arithmetic<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_BASE_SUM) gates::eval_base_sum<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_POSEIDON) poseidon_gate_dev(v, s);
    if constexpr (KIND == VPBS_GATE_POSEIDON_MDS) poseidon_mds_dev(v, s);
    if constexpr (KIND == VPBS_GATE_ARITHMETIC_EXT) gates::eval_arithmetic_ext<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_MUL_EXT) gates::eval_mul_ext<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_REDUCING) gates::eval_reducing<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_REDUCING_EXT) gates::eval_reducing_ext<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_RANDOM_ACCESS) gates::eval_random_access<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_EXPONENTIATION) gates::eval_exponentiation<u64>(g, v, s);
    if constexpr (KIND == VPBS_GATE_COSET_INTERPOLATION) gates::eval_coset_interpolation<u64>(g, t, v, s);
}

// grid: ceil(len / 256).  wires / consts: LDE columns with column stride `len` (the local leaves of the batches).
template <unsigned KIND, int NC>
__global__ void __launch_bounds__(THREADS)
gate_kernel(const u64* __restrict__ wires, const u64* __restrict__ consts, size_t len, size_t j0, size_t j1, vpbs_gate g,
            unsigned num_selectors, gates::CosetTables tables, const u64* __restrict__ apow, unsigned pow_stride, unsigned nc, PiHash pih,
            u64* __restrict__ out, int accumulate) {
    const size_t j = j0 + blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (j >= j1) return;
    DevVars v{wires, consts + (size_t)num_selectors * len, len, j, pih};
    DevSinkT<NC> s(apow, pow_stride, nc);
    eval_kind<KIND>(g, tables, v, s);
    const u64 filter = gates::compute_filter<u64>(g, consts[(size_t)g.selector_index * len + j], num_selectors > 1);
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        if ((unsigned)a >= nc) continue;
        u64 r = gl::mul(filter, s.acc[a].reduce());
        if (accumulate) r = gl::add(r, out[(size_t)a * len + j]);
        out[(size_t)a * len + j] = r;
    }
}

template <unsigned KIND>
void launch_one(hipStream_t s, const u64* wires, const u64* consts, size_t len, size_t j0, size_t j1, const vpbs_gate& g, unsigned num_selectors,
                const u64* apow, unsigned pow_stride, unsigned nc, const PiHash& pih, u64* out, int accumulate) {
    gates::CosetTables t{};
    if (KIND == VPBS_GATE_COSET_INTERPOLATION) t = gates::coset_tables(g.p0);
    const dim3 grid((unsigned)((j1 - j0 + THREADS - 1) / THREADS));
    if (nc <= 2)  // plonky2's num_challenges = 2: two accumulators keep the kernels at their natural register count
        hipLaunchKernelGGL((gate_kernel<KIND, 2>), grid, dim3(THREADS), 0, s, wires, consts, len, j0, j1, g, num_selectors, t, apow, pow_stride, nc,
                           pih, out, accumulate);
    else
        hipLaunchKernelGGL((gate_kernel<KIND, 4>), grid, dim3(THREADS), 0, s, wires, consts, len, j0, j1, g, num_selectors, t, apow, pow_stride, nc,
                           pih, out, accumulate);
}

// ---- all gates in ONE launch, the tile's columns staged in LDS (the default path) ----
// The per-gate launches above make every gate re-read its wires from HBM (the 135 wire columns are touched 7.6 times between them, and each
// launch read-modify-writes the output); a (tile x item) kernel that left the re-reads to the caches (rounds 2-4, removed in round 6: the
// counters showed 4.7 x the algorithmic bytes leaving the XCDs' L2 and the waves waiting on memory 43 % of their cycles) was the step between.
// Here a workgroup of eight waves owns 64 LDE points: it copies the
// 135 wire values and the selector / gate-constant values of those points into LDS ONCE (coalesced 512-byte rows, [column][64]), and
// after one barrier every wave evaluates ITS share of the gates for the same 64 points out of LDS (ds_read_b64 at constant offsets: no
// address arithmetic, ~100 cycles instead of a trip to L2 / HBM).  The shares are balanced by weight; the PoseidonGate -- a third of the
// work -- is cut into three independent pieces (poseidon_gate_part).  The waves' sums are added through LDS and ONE plane is written.
// HBM traffic = the algorithmic bytes; 73 KB of LDS per workgroup, two workgroups per CU = 4 waves per SIMD.
constexpr unsigned TILE_PTS = 64, TILE_WAVES = 8, TILE_MAX_COLS = 135 + 8, TILE_MAX_GATES = 20, TILE_MAX_UNITS = TILE_MAX_GATES + 2;
struct TilePlan {
    vpbs_gate gates[TILE_MAX_UNITS];   // work units ordered by wave
    uint8_t part[TILE_MAX_UNITS];      // 0: the whole gate; 1..3: a piece of the PoseidonGate
    gates::CosetTables tables;
    unsigned wave_first[TILE_WAVES + 1];
    unsigned n_wires, n_consts;        // columns staged: wires [0, n_wires), then constants_sigmas columns [0, n_consts)
};
using LdsWords = const __attribute__((address_space(3))) u64*;   // a pointer the compiler KNOWS is LDS: ds_read_b64 with the column as the
                                                                 // instruction's offset (a generic pointer reads through flat_load + 64-bit adds)
struct LdsVars {
    using F = u64;
    LdsWords lds;      // [n_wires + n_consts][64], this lane's column of it
    unsigned n_wires, first_const;   // first_const: LDS column of the first gate constant (n_wires + num_selectors)
    PiHash pih;
    __device__ __forceinline__ u64 wire(unsigned i) const { return lds[i * TILE_PTS]; }
    __device__ __forceinline__ u64 constant(unsigned i) const { return lds[(first_const + i) * TILE_PTS]; }
    __device__ __forceinline__ u64 pi_hash(unsigned i) const { return pih.h[i]; }
};
template <int NC>
__global__ void __launch_bounds__(TILE_PTS * TILE_WAVES, 4)
gate_tile_kernel(const u64* __restrict__ wires, const u64* __restrict__ consts, size_t len, TilePlan plan, unsigned num_selectors,
                 const u64* __restrict__ apow, unsigned pow_stride, unsigned nc, PiHash pih, u64* __restrict__ out,
                 unsigned long long* __restrict__ prof /* VPBS_TRACE_GATES: shader cycles per work unit, [TILE_MAX_UNITS]; else nullptr */) {
    __shared__ u64 lds[TILE_MAX_COLS * TILE_PTS];
    // the wave's number as a SCALAR (readfirstlane): its work units, their gates' kinds and parameters then live in scalar registers and the
    // dispatch over the kinds is a scalar branch -- as a vector value the compiler treated `u` as divergent: the gate descriptor came through
    // global_load into VGPRs and every case of the switch ran under exec masks with the sinks' accumulators merged by moves behind it
    const unsigned lane = threadIdx.x & (TILE_PTS - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TILE_PTS);
    const size_t j = (size_t)blockIdx.x * TILE_PTS + lane;
    // stage the tile: wave w copies columns w, w + 8, ... -- six 512-byte rows requested before the first is written to LDS, so that a
    // wave pays three memory round trips for its 18 columns, not eighteen
    {
        constexpr unsigned IN_FLIGHT = 6;
        const unsigned n_cols = plan.n_wires + plan.n_consts;
        for (unsigned c0 = wave; c0 < n_cols; c0 += TILE_WAVES * IN_FLIGHT) {
            u64 t[IN_FLIGHT];
#pragma unroll
            for (unsigned k = 0; k < IN_FLIGHT; ++k) {
                const unsigned c = c0 + TILE_WAVES * k;
                t[k] = c >= n_cols ? 0 : (c < plan.n_wires ? wires[(size_t)c * len + j] : consts[(size_t)(c - plan.n_wires) * len + j]);
            }
#pragma unroll
            for (unsigned k = 0; k < IN_FLIGHT; ++k) {
                const unsigned c = c0 + TILE_WAVES * k;
                if (c < n_cols) lds[c * TILE_PTS + lane] = t[k];
            }
        }
    }
    __syncthreads();
    u64 total[NC];
#pragma unroll
    for (int a = 0; a < NC; ++a) total[a] = 0;
    for (unsigned u = plan.wave_first[wave]; u < plan.wave_first[wave + 1]; ++u) {
        const vpbs_gate& g = plan.gates[u];
        const unsigned long long t_unit = prof ? __builtin_amdgcn_s_memtime() : 0;
        unsigned lds_at = (unsigned)(uintptr_t)(LdsWords)(lds + lane);   // the 32-bit LDS address of this lane's column
        asm volatile("" : "+v"(lds_at));   // opaque per unit: keeps the cases' LDS addresses from being hoisted and kept live across all of them
        const LdsWords base = (LdsWords)(uintptr_t)lds_at;
        LdsVars v{base, plan.n_wires, plan.n_wires + num_selectors, pih};
        DevSinkT<NC> s(apow, pow_stride, nc);
        if (g.kind == VPBS_GATE_POSEIDON && plan.part[u]) {
            switch (plan.part[u]) {
                case 1: poseidon_gate_part<1>(v, s); break;
                case 2: poseidon_gate_part<2>(v, s); break;
                default: poseidon_gate_part<3>(v, s); break;
            }
        } else {
            switch (g.kind) {  // wave-uniform
#define VPBS_TILE_CASE(K) case K: eval_kind<K>(g, plan.tables, v, s); break;
                VPBS_TILE_CASE(VPBS_GATE_CONSTANT)
                VPBS_TILE_CASE(VPBS_GATE_PUBLIC_INPUT)
                VPBS_TILE_CASE(VPBS_GATE_ARITHMETIC)
                VPBS_TILE_CASE(VPBS_GATE_BASE_SUM)
                VPBS_TILE_CASE(VPBS_GATE_POSEIDON)
                VPBS_TILE_CASE(VPBS_GATE_POSEIDON_MDS)
                VPBS_TILE_CASE(VPBS_GATE_ARITHMETIC_EXT)
                VPBS_TILE_CASE(VPBS_GATE_MUL_EXT)
                VPBS_TILE_CASE(VPBS_GATE_REDUCING)
                VPBS_TILE_CASE(VPBS_GATE_REDUCING_EXT)
                VPBS_TILE_CASE(VPBS_GATE_RANDOM_ACCESS)
                VPBS_TILE_CASE(VPBS_GATE_EXPONENTIATION)
                VPBS_TILE_CASE(VPBS_GATE_COSET_INTERPOLATION)
#undef VPBS_TILE_CASE
                default: break;
            }
        }
        const u64 filter = gates::compute_filter<u64>(g, base[(plan.n_wires + g.selector_index) * TILE_PTS], num_selectors > 1);
#pragma unroll
        for (int a = 0; a < NC; ++a)
            if ((unsigned)a < nc) total[a] = gl::add(total[a], gl::mul(filter, s.acc[a].reduce()));
        if (prof && lane == 0) atomicAdd(prof + u, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_unit) + (total[0] & 0));
    }
    // the waves' sums: through LDS (the tile is not needed any more), added by the first nc waves
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NC; ++a)
        if ((unsigned)a < nc) lds[(wave * NC + a) * TILE_PTS + lane] = total[a];
    __syncthreads();
    if (wave < nc) {
        u64 r = lds[wave * TILE_PTS + lane];   // wave 0's sum for challenge `wave`
        for (unsigned w = 1; w < TILE_WAVES; ++w) r = gl::add(r, lds[(w * NC + wave) * TILE_PTS + lane]);
        out[(size_t)wave * len + j] = r;
    }
}

__global__ void sum_planes_kernel(const u64* __restrict__ planes, unsigned n_planes, size_t words, u64* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= words) return;
    u64 r = planes[i];
    for (unsigned p = 1; p < n_planes; ++p) r = gl::add(r, planes[(size_t)p * words + i]);
    out[i] = r;
}

// ---- host: Gate::id() strings, derived parameters, sorting ----
const char* const FIELD = "plonky2_field::goldilocks_field::GoldilocksField";
std::string gate_id(const vpbs_gate& g) {
    auto u = [](unsigned x) { return std::to_string(x); };
    switch (g.kind) {
        case VPBS_GATE_NOOP: return "NoopGate";
        case VPBS_GATE_CONSTANT: return "ConstantGate { num_consts: " + u(g.p0) + " }";
        case VPBS_GATE_PUBLIC_INPUT: return "PublicInputGate";
        case VPBS_GATE_ARITHMETIC: return "ArithmeticGate { num_ops: " + u(g.p0) + " }";
        case VPBS_GATE_BASE_SUM: return "BaseSumGate { num_limbs: " + u(g.p0) + " } + Base: " + u(g.p1);
        case VPBS_GATE_POSEIDON: return std::string("PoseidonGate(PhantomData<") + FIELD + ">)<WIDTH=12>";
        case VPBS_GATE_POSEIDON_MDS: return std::string("PoseidonMdsGate(PhantomData<") + FIELD + ">)<WIDTH=12>";
        case VPBS_GATE_ARITHMETIC_EXT: return "ArithmeticExtensionGate { num_ops: " + u(g.p0) + " }";
        case VPBS_GATE_MUL_EXT: return "MulExtensionGate { num_ops: " + u(g.p0) + " }";
        case VPBS_GATE_REDUCING: return "ReducingGate { num_coeffs: " + u(g.p0) + " }";
        case VPBS_GATE_REDUCING_EXT: return "ReducingExtensionGate { num_coeffs: " + u(g.p0) + " }";
        case VPBS_GATE_RANDOM_ACCESS:
            return "RandomAccessGate { bits: " + u(g.p0) + ", num_copies: " + u(g.p1) + ", num_extra_constants: " + u(g.p2) +
                   ", _phantom: PhantomData<" + FIELD + "> }<D=2>";
        case VPBS_GATE_EXPONENTIATION:
            return "ExponentiationGate { num_power_bits: " + u(g.p0) + ", _phantom: PhantomData<" + FIELD + "> }<D=2>";
        case VPBS_GATE_COSET_INTERPOLATION: {
            // Debug of the struct prints the barycentric weights too; they are a function of subgroup_bits
            const gates::CosetTables t = gates::coset_tables(g.p0);
            std::string w;
            for (unsigned i = 0; i < (1u << g.p0); ++i) w += (i ? ", " : "") + std::to_string(t.weights[i]);
            return "CosetInterpolationGate { subgroup_bits: " + u(g.p0) + ", degree: " + u(g.p1) + ", barycentric_weights: [" + w +
                   "], _phantom: PhantomData<" + FIELD + "> }<D=2>";
        }
        default: return "";
    }
}

constexpr unsigned CFG_WIRES = 135, CFG_ROUTED = 80, CFG_CONSTANTS = 2, CFG_MAX_DEGREE = 8;  // standard_recursion_config

// derived fields; false when the parameters are not supported
bool derive(vpbs_gate& g) {
    switch (g.kind) {
        case VPBS_GATE_NOOP: g.degree = 0; g.num_constraints = 0; g.num_constants = 0; g.num_wires = 0; return true;
        case VPBS_GATE_CONSTANT: g.degree = 1; g.num_constraints = g.p0; g.num_constants = g.p0; g.num_wires = g.p0; return g.p0 >= 1;
        case VPBS_GATE_PUBLIC_INPUT: g.degree = 1; g.num_constraints = 4; g.num_constants = 0; g.num_wires = 4; return true;
        case VPBS_GATE_ARITHMETIC: g.degree = 3; g.num_constraints = g.p0; g.num_constants = 2; g.num_wires = 4 * g.p0; return g.p0 >= 1;
        case VPBS_GATE_BASE_SUM:
            g.degree = g.p1; g.num_constraints = 1 + g.p0; g.num_constants = 0; g.num_wires = 1 + g.p0;
            return g.p0 >= 1 && g.p1 >= 2 && g.p1 <= 7;
        case VPBS_GATE_POSEIDON: g.degree = 7; g.num_constraints = 123; g.num_constants = 0; g.num_wires = 135; return true;
        case VPBS_GATE_POSEIDON_MDS: g.degree = 1; g.num_constraints = 24; g.num_constants = 0; g.num_wires = 48; return true;
        case VPBS_GATE_ARITHMETIC_EXT: g.degree = 3; g.num_constraints = 2 * g.p0; g.num_constants = 2; g.num_wires = 8 * g.p0; return g.p0 >= 1;
        case VPBS_GATE_MUL_EXT: g.degree = 3; g.num_constraints = 2 * g.p0; g.num_constants = 1; g.num_wires = 6 * g.p0; return g.p0 >= 1;
        case VPBS_GATE_REDUCING: g.degree = 2; g.num_constraints = 2 * g.p0; g.num_constants = 0; g.num_wires = 3 * g.p0 + 4; return g.p0 >= 1;
        case VPBS_GATE_REDUCING_EXT: g.degree = 2; g.num_constraints = 2 * g.p0; g.num_constants = 0; g.num_wires = 4 * g.p0 + 4; return g.p0 >= 1;
        case VPBS_GATE_RANDOM_ACCESS:
            g.degree = g.p0 + 1; g.num_constraints = g.p1 * (g.p0 + 2) + g.p2; g.num_constants = g.p2;
            g.num_wires = (2 + (1u << g.p0)) * g.p1 + g.p2 + g.p1 * g.p0;
            return g.p0 >= 1 && g.p0 <= 5 && g.p1 >= 1;
        case VPBS_GATE_EXPONENTIATION: g.degree = 4; g.num_constraints = g.p0 + 1; g.num_constants = 0; g.num_wires = 2 + 2 * g.p0; return g.p0 >= 1;
        case VPBS_GATE_COSET_INTERPOLATION: {
            if (g.p0 < 1 || g.p0 > 5 || g.p1 < 2) return false;
            const unsigned points = 1u << g.p0, ni = (points - 2) / (g.p1 - 1);
            g.degree = g.p1; g.num_constraints = 2 * (2 + 2 * ni); g.num_constants = 0; g.num_wires = 1 + 2 * points + 4 + 4 * ni + 2;
            return true;
        }
        default: return false;
    }
}

// ---- host evaluation over GF(p^2) (verifier) ----
struct HostVars {
    using F = gl::Ext;
    const u64* wires;      // [n_wires][2]
    const u64* constants;  // first gate constant, [..][2]
    unsigned n_wires, n_constants;
    const u64* pih;
    gl::Ext wire(unsigned i) const { return i < n_wires ? gl::Ext{wires[2 * i], wires[2 * i + 1]} : gl::ext(0); }
    gl::Ext constant(unsigned i) const { return i < n_constants ? gl::Ext{constants[2 * i], constants[2 * i + 1]} : gl::ext(0); }
    u64 pi_hash(unsigned i) const { return pih[i]; }
};
struct HostSink {
    std::vector<gl::Ext> c;
    void push(gl::Ext x) { c.push_back(x); }
};

}  // namespace

namespace {
// one gate's kernel on stream s, into out (accumulate = 0: overwrite)
bool launch_gate(hipStream_t s, const u64* wires_lde, const u64* consts_lde, size_t len, const vpbs_gate& g, unsigned num_selectors,
                 const u64* d_apow, unsigned pow_stride, unsigned nc, const PiHash& pih, u64* out, int accumulate) {
#define VPBS_GATE_CASE(K) \
    case K: launch_one<K>(s, wires_lde, consts_lde, len, 0, len, g, num_selectors, d_apow, pow_stride, nc, pih, out, accumulate); return true;
    switch (g.kind) {
        VPBS_GATE_CASE(VPBS_GATE_CONSTANT)
        VPBS_GATE_CASE(VPBS_GATE_PUBLIC_INPUT)
        VPBS_GATE_CASE(VPBS_GATE_ARITHMETIC)
        VPBS_GATE_CASE(VPBS_GATE_BASE_SUM)
        VPBS_GATE_CASE(VPBS_GATE_POSEIDON)
        VPBS_GATE_CASE(VPBS_GATE_POSEIDON_MDS)
        VPBS_GATE_CASE(VPBS_GATE_ARITHMETIC_EXT)
        VPBS_GATE_CASE(VPBS_GATE_MUL_EXT)
        VPBS_GATE_CASE(VPBS_GATE_REDUCING)
        VPBS_GATE_CASE(VPBS_GATE_REDUCING_EXT)
        VPBS_GATE_CASE(VPBS_GATE_RANDOM_ACCESS)
        VPBS_GATE_CASE(VPBS_GATE_EXPONENTIATION)
        VPBS_GATE_CASE(VPBS_GATE_COSET_INTERPOLATION)
        default: return false;
    }
#undef VPBS_GATE_CASE
}
// measured kernel time per gate at 2^18 points (us): the weights of the lane assignment below
unsigned gate_weight(const vpbs_gate& g) {
    switch (g.kind) {
        case VPBS_GATE_POSEIDON: return 180;
        case VPBS_GATE_CONSTANT:
        case VPBS_GATE_PUBLIC_INPUT: return 9;
        case VPBS_GATE_COSET_INTERPOLATION: return 55;
        case VPBS_GATE_BASE_SUM: return 58;
        default: return 8 + g.num_wires / 2;  // the HBM-bound gates: time follows the number of wire columns read
    }
}
__global__ void add_lanes_kernel(u64* __restrict__ out, const u64* __restrict__ a, const u64* __restrict__ b, size_t words) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= words) return;
    u64 r = out[i];
    if (a) r = gl::add(r, a[i]);
    if (b) r = gl::add(r, b[i]);
    out[i] = r;
}
}  // namespace

// (Splitting the points into row chunks so that one chunk's columns stay in the 256 MB memory-side cache across the 13
// kernels was measured and is slower at every chunk count: the kernels need the whole 2^18-point grid to hide latency.)
//
// lanes: the gates are spread over up to three streams (the caller's + two helpers, forked and joined with events) so that the
// VALU-bound PoseidonGate kernel overlaps the HBM-bound ones instead of queueing behind them; each lane accumulates into its own
// buffer (lane_out[1], lane_out[2]: [nc][len] scratch, may be null = single lane) and the lanes are summed at the join.
void launch_gate_terms(hipStream_t s, const u64* wires_lde, const u64* consts_lde, size_t len, const vpbs_gate* gs, unsigned n_gates,
                       unsigned num_selectors, const u64 pi_hash[4], const u64* d_apow, unsigned pow_stride, unsigned nc, u64* d_out,
                       GateLanes* lanes) {
    PiHash pih{{pi_hash[0], pi_hash[1], pi_hash[2], pi_hash[3]}};
    const unsigned n_lanes = lanes ? 3 : 1;
    // longest-processing-time-first assignment
    std::vector<unsigned> order;
    for (unsigned i = 0; i < n_gates; ++i)
        if (gs[i].num_constraints) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return gate_weight(gs[a]) > gate_weight(gs[b]); });
    std::vector<unsigned> lane_of(n_gates, 0);
    unsigned load[3] = {0, 0, 0};
    unsigned extra_lane = 0;
    if (lanes && lanes->extra) {  // the heaviest item after the Poseidon gate: placed first on the emptiest helper lane
        extra_lane = 1;
        load[1] += lanes->extra_weight;
    }
    for (unsigned i : order) {
        unsigned best = 0;
        for (unsigned l = 1; l < n_lanes; ++l)
            if (load[l] < load[best]) best = l;
        lane_of[i] = best;
        load[best] += gate_weight(gs[i]);
    }
    hipStream_t st[3] = {s, lanes ? lanes->stream[0] : nullptr, lanes ? lanes->stream[1] : nullptr};
    u64* out[3] = {d_out, lanes ? lanes->out[0] : nullptr, lanes ? lanes->out[1] : nullptr};
    if (n_lanes > 1) {
        (void)hipEventRecord(lanes->fork, s);
        for (unsigned l = 1; l < 3; ++l) (void)hipStreamWaitEvent(st[l], lanes->fork, 0);
    }
    bool used[3] = {false, false, false};
    if (lanes && lanes->extra) lanes->extra(st[extra_lane], lanes->extra_arg);
    for (unsigned l = 0; l < n_lanes; ++l)
        for (unsigned i : order)
            if (lane_of[i] == l && launch_gate(st[l], wires_lde, consts_lde, len, gs[i], num_selectors, d_apow, pow_stride, nc, pih, out[l], used[l] ? 1 : 0))
                used[l] = true;
    if (!used[0] && !(lanes && lanes->skip_sum)) (void)hipMemsetAsync(d_out, 0, sizeof(u64) * nc * len, s);
    if (n_lanes > 1) {
        for (unsigned l = 1; l < 3; ++l) {
            (void)hipEventRecord(lanes->join[l - 1], st[l]);
            (void)hipStreamWaitEvent(s, lanes->join[l - 1], 0);
        }
        for (unsigned l = 0; l < 3; ++l) lanes->used[l] = used[l];
        if (!lanes->skip_sum && (used[1] || used[2])) {
            const size_t words = (size_t)nc * len;
            hipLaunchKernelGGL(add_lanes_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, d_out, used[1] ? out[1] : (const u64*)nullptr,
                               used[2] ? out[2] : (const u64*)nullptr, words);
        }
    }
}

// Work units of the LDS-tile kernel: every gate with constraints, the PoseidonGate as three pieces, packed heaviest first into the eight
// waves of a workgroup.  False when the gate set does not fit (too many gates, two CosetInterpolationGates, more constant columns than
// the tile holds).
// Relative cost of a gate's constraints in the tile kernel: microseconds of a launch over 2^19 points in which ONE wave per workgroup
// evaluates that gate alone (tools/time_gates.py, staging time subtracted) -- a lone wave runs at its dependent-instruction latency, so the
// figure follows the instruction count (~0.07 per instruction) -- scaled by the gate's parameters.  TILE_UNIT_COST: what every work unit
// carried besides when the weights were taken (selector filter, the reductions of the lazily folded sums, dispatch: ~700 instructions then;
// ~150 since the single 160-bit reduction and the branch-free sink -- the weights are relative, the balance they give was re-measured: the
// kernel alone runs at 0.93 of its instruction time).  (The s_memtime spans that
// VPBS_TRACE_GATES prints are NOT such a measure: a wave's span stretches with whatever shares its SIMD.)
constexpr double TILE_UNIT_COST = 50;
static double tile_weight(const vpbs_gate& g) {
    switch (g.kind) {
        case VPBS_GATE_ARITHMETIC: return 8.0 * g.p0;
        case VPBS_GATE_BASE_SUM: return 8.0 * g.p0 * (g.p1 - 1);
        case VPBS_GATE_POSEIDON_MDS: return 60;
        case VPBS_GATE_ARITHMETIC_EXT: return 17.0 * g.p0;
        case VPBS_GATE_MUL_EXT: return 13.5 * g.p0;
        case VPBS_GATE_REDUCING: return 10.5 * g.p0;
        case VPBS_GATE_REDUCING_EXT: return 11.7 * g.p0;
        case VPBS_GATE_RANDOM_ACCESS: return 235.0 * g.p1 * (1u << g.p0) / 64;
        case VPBS_GATE_EXPONENTIATION: return 8.0 * g.p0;
        case VPBS_GATE_COSET_INTERPOLATION: return 13.4 * (1u << g.p0);
        default: return 0.5 * g.num_constraints;
    }
}
static bool make_tile_plan(const vpbs_gate* gs, unsigned n_gates, unsigned num_selectors, TilePlan& plan) {
    struct Unit {
        unsigned gate, part;
        double weight;
    };
    std::vector<Unit> units;
    unsigned n_coset = 0, max_consts = 0, n_wires = 0;
    for (unsigned i = 0; i < n_gates; ++i) {
        if (!gs[i].num_constraints) continue;
        max_consts = std::max(max_consts, gs[i].num_constants);
        n_wires = std::max(n_wires, gs[i].num_wires);
        if (gs[i].kind == VPBS_GATE_POSEIDON) {   // the three pieces: 7.5 k / 4.7 k / 6.0 k of the gate's 18 k instructions
            units.push_back({i, 1, 510 + TILE_UNIT_COST});
            units.push_back({i, 2, 298 + TILE_UNIT_COST});
            units.push_back({i, 3, 404 + TILE_UNIT_COST});
            continue;
        }
        if (gs[i].kind == VPBS_GATE_COSET_INTERPOLATION) {
            if (n_coset++) return false;
            plan.tables = gates::coset_tables(gs[i].p0);
        }
        units.push_back({i, 0, tile_weight(gs[i]) + TILE_UNIT_COST});
    }
    if (units.empty() || units.size() > TILE_MAX_UNITS || n_wires > 135 || n_wires + num_selectors + max_consts > TILE_MAX_COLS) return false;
    // heaviest first onto the lightest wave.  (Cutting the loop gates -- BaseSum, Reducing, ... -- into iteration ranges to level the
    // waves exactly was built and measured: the run-time loop bounds cost the evaluators 6 % and every extra unit its fixed cost, more than
    // the better balance returned: 995-1050 us against 939 us for the cyclic circuit's gate set.)
    std::sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.weight > b.weight; });
    std::vector<std::vector<Unit>> bins(TILE_WAVES);
    double load[TILE_WAVES] = {0};
    for (const Unit& u : units) {
        const unsigned b = (unsigned)(std::min_element(load, load + TILE_WAVES) - load);
        bins[b].push_back(u);
        load[b] += u.weight;
    }
    unsigned k = 0;
    for (unsigned w = 0; w < TILE_WAVES; ++w) {
        plan.wave_first[w] = k;
        for (const Unit& u : bins[w]) {
            plan.gates[k] = gs[u.gate];
            plan.part[k] = (uint8_t)u.part;
            ++k;
        }
    }
    plan.wave_first[TILE_WAVES] = k;
    plan.n_wires = n_wires;
    plan.n_consts = num_selectors + max_consts;
    static const bool trace = getenv("VPBS_TRACE_GATES") != nullptr;
    if (trace) {
        for (unsigned w = 0; w < TILE_WAVES; ++w) {
            std::fprintf(stderr, "[gate tile] wave %u load %.0f:", w, load[w]);
            for (const Unit& u : bins[w]) std::fprintf(stderr, " kind %u part %u", gs[u.gate].kind, u.part);
            std::fprintf(stderr, "\n");
        }
    }
    return true;
}

unsigned gate_terms_planes(const vpbs_gate* gs, unsigned n_gates, unsigned num_selectors, const Tuning& tune, size_t len) {
    if (tune.gates_tile && len % TILE_PTS == 0) {
        TilePlan tp{};
        if (make_tile_plan(gs, n_gates, num_selectors, tp)) return 1;
    }
    return 0;   // the gate set does not fit a tile plan: the caller takes the per-gate launches
}

// d_planes: [1][nc][len]; returns the number of planes written (0: the gate set does not fit a tile plan or the tile kernel is switched off, nothing launched)
unsigned launch_gate_terms_fused(hipStream_t s, const Tuning& tune, const u64* wires_lde, const u64* consts_lde, size_t len, const vpbs_gate* gs,
                                 unsigned n_gates, unsigned num_selectors, const u64 pi_hash[4], const u64* d_apow, unsigned pow_stride, unsigned nc,
                                 u64* d_planes) {
    PiHash pih{{pi_hash[0], pi_hash[1], pi_hash[2], pi_hash[3]}};
    if (tune.gates_tile && len % TILE_PTS == 0) {
        TilePlan tp{};
        if (make_tile_plan(gs, n_gates, num_selectors, tp)) {
            const dim3 grid((unsigned)(len / TILE_PTS)), block(TILE_PTS * TILE_WAVES);
            static const bool trace = getenv("VPBS_TRACE_GATES") != nullptr;
            unsigned long long* prof = nullptr;
            if (trace) {   // development aid: the shader cycles every work unit really takes inside the loaded kernel (what tile_weight approximates)
                (void)hipMalloc(&prof, sizeof(unsigned long long) * TILE_MAX_UNITS);
                (void)hipMemsetAsync(prof, 0, sizeof(unsigned long long) * TILE_MAX_UNITS, s);
            }
            // NC = nc exactly: the branch-free sink accumulates NC sums per constraint whether they are read or not (one or three challenges
            // under NC = 2 / 4 paid eight multiply-adds per constraint for a sum nobody reads: ADVICE r05).  nc = 0 never gets here (callers check).
            if (nc < 1 || nc > 4) return 0;
#define VPBS_TILE_LAUNCH(NC_) \
    hipLaunchKernelGGL((gate_tile_kernel<NC_>), grid, block, 0, s, wires_lde, consts_lde, len, tp, num_selectors, d_apow, pow_stride, nc, pih, d_planes, prof)
            switch (nc) {
                case 1: VPBS_TILE_LAUNCH(1); break;
                case 2: VPBS_TILE_LAUNCH(2); break;
                case 3: VPBS_TILE_LAUNCH(3); break;
                default: VPBS_TILE_LAUNCH(4); break;
            }
#undef VPBS_TILE_LAUNCH
            if (trace) {
                unsigned long long h[TILE_MAX_UNITS];
                (void)vpbs::stream_sync(s);
                (void)hipMemcpy(h, prof, sizeof h, hipMemcpyDeviceToHost);
                (void)hipFree(prof);
                for (unsigned w = 0; w < TILE_WAVES; ++w) {
                    unsigned long long sum = 0;
                    std::fprintf(stderr, "[gate tile cycles] wave %u:", w);
                    for (unsigned u = tp.wave_first[w]; u < tp.wave_first[w + 1]; ++u) {
                        std::fprintf(stderr, " kind %u part %u %.0f", tp.gates[u].kind, tp.part[u], (double)h[u] / grid.x);
                        sum += h[u];
                    }
                    std::fprintf(stderr, "  total %.0f\n", (double)sum / grid.x);
                }
            }
            return 1;
        }
    }
    return 0;
}

void launch_sum_planes(hipStream_t s, const u64* d_planes, unsigned n_planes, size_t words, u64* d_out) {
    hipLaunchKernelGGL(sum_planes_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, d_planes, n_planes, words, d_out);
}

// checks a laid-out gate list against the batches it will be evaluated on
void validate_gates(const vpbs_gate* gs, unsigned n_gates, unsigned num_selectors, unsigned n_constants_cols, unsigned n_wires) {
    VPBS_REQUIRE(gs && n_gates >= 1 && num_selectors >= 1, "no gates");
    for (unsigned i = 0; i < n_gates; ++i) {
        vpbs_gate g = gs[i];
        VPBS_REQUIRE(derive(g), "unsupported gate parameters");
        VPBS_REQUIRE(g.degree == gs[i].degree && g.num_constraints == gs[i].num_constraints && g.num_wires == gs[i].num_wires,
                     "gate list was not produced by vpbs_gates_layout");
        VPBS_REQUIRE(g.num_wires <= n_wires, "gate needs more wires than the wires batch has");
        VPBS_REQUIRE(gs[i].selector_index < num_selectors && num_selectors + g.num_constants <= n_constants_cols,
                     "selector / gate-constant columns out of range");
        VPBS_REQUIRE(gs[i].group_start <= gs[i].index && gs[i].index < gs[i].group_end && gs[i].group_end - gs[i].group_start <= 16,
                     "bad selector group");
    }
}

// sum_i alpha^i sum_g filter_g c_{g,i} at one extension point
void gate_terms_at(const vpbs_gate* gs, unsigned n_gates, unsigned num_selectors, const u64* constants_at, unsigned n_constants,
                   const u64* wires_at, unsigned n_wires, const u64 pi_hash[4], const u64* alphas, unsigned nc, u64* out) {
    unsigned max_c = 0;
    for (unsigned i = 0; i < n_gates; ++i) max_c = std::max(max_c, gs[i].num_constraints);
    std::vector<gl::Ext> total(max_c, gl::ext(0));
    for (unsigned i = 0; i < n_gates; ++i) {
        const vpbs_gate& g = gs[i];
        HostVars v{wires_at, constants_at + 2 * (size_t)num_selectors, n_wires, n_constants - num_selectors, pi_hash};
        HostSink s;
        gates::CosetTables t{};
        if (g.kind == VPBS_GATE_COSET_INTERPOLATION) t = gates::coset_tables(g.p0);
        gates::eval_gate<gl::Ext>(g, &t, v, s);
        const gl::Ext sel{constants_at[2 * (size_t)g.selector_index], constants_at[2 * (size_t)g.selector_index + 1]};
        const gl::Ext filter = gates::compute_filter<gl::Ext>(g, sel, num_selectors > 1);
        for (size_t k = 0; k < s.c.size() && k < total.size(); ++k) total[k] = gl::add(total[k], gl::mul(filter, s.c[k]));
    }
    for (unsigned a = 0; a < nc; ++a) {
        gl::Ext acc = gl::ext(0);
        for (size_t k = total.size(); k-- > 0;) acc = gl::add(gl::mul(acc, alphas[a]), total[k]);
        out[2 * a] = acc.c0;
        out[2 * a + 1] = acc.c1;
    }
}
}  // namespace vpbs

extern "C" {

int vpbs_gate_default_params(vpbs_gate* g) {
    if (!g) return VPBS_ERR_INVALID;
    using namespace vpbs;
    switch (g->kind) {
        case VPBS_GATE_CONSTANT: if (!g->p0) g->p0 = CFG_CONSTANTS; break;
        case VPBS_GATE_ARITHMETIC: if (!g->p0) g->p0 = CFG_ROUTED / 4; break;                 // ArithmeticGate::new_from_config
        case VPBS_GATE_BASE_SUM:
            if (!g->p1) g->p1 = 2;
            if (!g->p0) {  // BaseSumGate::new_from_config: min(floor(log_B(p)), num_routed_wires - 1)
                unsigned l = 0;
                for (unsigned __int128 v = g->p1; v <= (unsigned __int128)gl::P; v *= g->p1) ++l;
                g->p0 = std::min(l, CFG_ROUTED - 1);
            }
            break;
        case VPBS_GATE_ARITHMETIC_EXT: if (!g->p0) g->p0 = CFG_ROUTED / 8; break;
        case VPBS_GATE_MUL_EXT: if (!g->p0) g->p0 = CFG_ROUTED / 6; break;
        case VPBS_GATE_REDUCING: if (!g->p0) g->p0 = std::min(CFG_ROUTED - 6, (CFG_WIRES - 4) / 3); break;      // max_coeffs_len
        case VPBS_GATE_REDUCING_EXT: if (!g->p0) g->p0 = std::min((CFG_ROUTED - 6) / 2, (CFG_WIRES - 4) / 4); break;
        case VPBS_GATE_RANDOM_ACCESS: {
            if (!g->p0) g->p0 = 4;
            if (g->p0 > 5) return VPBS_ERR_INVALID;
            if (!g->p1) {  // RandomAccessGate::new_from_config
                const unsigned vec = 1u << g->p0;
                g->p1 = std::min(CFG_ROUTED / (2 + vec), CFG_WIRES / (2 + vec + g->p0));
                g->p2 = std::min(CFG_ROUTED - (2 + vec) * g->p1, CFG_CONSTANTS);
            }
            break;
        }
        case VPBS_GATE_EXPONENTIATION: if (!g->p0) g->p0 = std::min(CFG_ROUTED - 2, (CFG_WIRES - 2) / 2); break;
        case VPBS_GATE_COSET_INTERPOLATION: {
            if (!g->p0) g->p0 = 4;
            if (g->p0 > 5) return VPBS_ERR_INVALID;
            if (!g->p1) {  // CosetInterpolationGate::with_max_degree(subgroup_bits, max_degree)
                const unsigned points = 1u << g->p0;
                const unsigned ni = (points - 2) / (CFG_MAX_DEGREE - 1);
                g->p1 = (points - 2) / (ni + 1) + 2;
            }
            break;
        }
        default: break;
    }
    return vpbs::derive(*g) ? VPBS_OK : VPBS_ERR_INVALID;
}

int vpbs_gate_id(const vpbs_gate* g, char* buf, size_t len) {
    if (!g || !buf || !len) return VPBS_ERR_INVALID;
    const std::string id = vpbs::gate_id(*g);
    if (id.empty() || id.size() + 1 > len) return VPBS_ERR_INVALID;
    std::memcpy(buf, id.c_str(), id.size() + 1);
    return (int)id.size();
}

int vpbs_gates_layout(vpbs_gate* gs, unsigned n_gates, unsigned max_degree, unsigned* num_selectors, unsigned* num_gate_constraints) {
    if (!gs || !n_gates || !num_selectors || max_degree < 2) return VPBS_ERR_INVALID;
    std::vector<std::pair<std::string, vpbs_gate>> v;
    for (unsigned i = 0; i < n_gates; ++i) {
        if (!vpbs::derive(gs[i])) return VPBS_ERR_INVALID;
        v.push_back({vpbs::gate_id(gs[i]), gs[i]});
    }
    // CircuitBuilder::build: gates.sort_unstable_by_key(|g| (g.0.degree(), g.0.id()))
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) {
        return a.second.degree != b.second.degree ? a.second.degree < b.second.degree : a.first < b.first;
    });
    for (unsigned i = 1; i < n_gates; ++i)
        if (v[i].first == v[i - 1].first) return VPBS_ERR_INVALID;  // a gate type appears once in a circuit's gate set
    unsigned max_c = 0;
    for (unsigned i = 0; i < n_gates; ++i) {
        gs[i] = v[i].second;
        gs[i].index = i;
        max_c = std::max(max_c, gs[i].num_constraints);
    }
    const unsigned max_gate_degree = gs[n_gates - 1].degree;
    if (max_gate_degree + n_gates - 1 <= max_degree) {  // selector_polynomials: one selector is enough
        for (unsigned i = 0; i < n_gates; ++i) {
            gs[i].selector_index = 0;
            gs[i].group_start = 0;
            gs[i].group_end = n_gates;
        }
        *num_selectors = 1;
    } else {
        if (max_gate_degree >= max_degree) return VPBS_ERR_INVALID;  // "No gate can have degree >= max_degree"
        unsigned start = 0, groups = 0;
        while (start < n_gates) {
            unsigned size = 0;
            while (start + size < n_gates && size + gs[start + size].degree < max_degree) ++size;
            for (unsigned i = start; i < start + size; ++i) {
                gs[i].selector_index = groups;
                gs[i].group_start = start;
                gs[i].group_end = start + size;
            }
            start += size;
            ++groups;
        }
        *num_selectors = groups;
    }
    if (num_gate_constraints) *num_gate_constraints = max_c;
    return VPBS_OK;
}

int vpbs_gate_terms_at(const vpbs_gate* gs, unsigned n_gates, unsigned num_selectors, const uint64_t* constants_at, unsigned n_constants,
                       const uint64_t* wires_at, unsigned n_wires, const uint64_t pi_hash[4], const uint64_t* alphas, unsigned nc,
                       uint64_t* out) {
    if (!gs || !n_gates || !constants_at || !wires_at || !pi_hash || !alphas || !out || num_selectors > n_constants || !nc)
        return VPBS_ERR_INVALID;
    try {
        vpbs::validate_gates(gs, n_gates, num_selectors, n_constants, n_wires);
    } catch (const vpbs::DeviceError&) {
        return VPBS_ERR_INVALID;
    }
    vpbs::gate_terms_at(gs, n_gates, num_selectors, constants_at, n_constants, wires_at, n_wires, pi_hash, alphas, nc, out);
    return VPBS_OK;
}

}  // extern "C"
