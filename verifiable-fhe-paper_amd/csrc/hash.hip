// Batched Poseidon sponge over LDE rows, Merkle levels + cap, proof-of-work search, for gfx950.
// Replaces plonky2 0.2.0 hash/merkle_tree.rs `MerkleTree::new` (leaf digests = H::hash_or_noop, parents =
// H::two_to_one), hash/hashing.rs `hash_n_to_m_no_pad` (overwrite-mode sponge, rate 8) and fri/prover.rs
// `fri_proof_of_work`, reached from prove() at /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364
// (SURVEY.md 8a rows a5/a6/a11).  One lane = one leaf / node / nonce: the column-major, leaf-ordered LDE makes every
// absorb a fully coalesced 512-B wave read.  Integer-VALU bound (~15.3k instructions per permutation); no MFMA.
#include <cstdlib>

#include <algorithm>
#include "kernels.h"
#include "poseidon.h"

namespace vpbs {
namespace {
constexpr unsigned THREADS = 256;

// SAMPLE: the timing form (bench.py's in-kernel clock) -- the same code with two counters read at both ends; the form the prover launches
// carries neither the pointer nor the two 64-bit start values through its 15 k instructions (seven scalar registers the sponge loop wants)
template <bool SAMPLE>
__global__ void __launch_bounds__(THREADS)
leaf_hash_kernel(const u64* __restrict__ lde, unsigned ncols, size_t n_leaves, size_t col_stride, u64* __restrict__ digests,
                 u64* __restrict__ clock_sample) {
    const size_t j = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (j >= n_leaves) return;
    // timing runs only: one wave in the middle of the grid reports the shader cycles and the 100 MHz ticks of its own lifetime -- the clock the
    // chip really sustains under this kernel (a light probe kernel reads the boost clock instead)
    const bool sample = SAMPLE && clock_sample != nullptr && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
    u64 c0 = 0, r0 = 0;
    if (sample) {
        r0 = __builtin_amdgcn_s_memrealtime();
        c0 = __builtin_amdgcn_s_memtime();
    }
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = 0;
    if (ncols <= 4) {  // hash_or_noop: leaves of <= 4 elements are padded, not hashed
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((unsigned)k < ncols) s[k] = lde[k * col_stride + j];
    } else {
        // the next block's eight column values are requested before the current permutation (~15 k instructions) so that no
        // wave ever waits on HBM in front of a permutation
        u64 nxt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) nxt[k] = (unsigned)k < ncols ? lde[(size_t)k * col_stride + j] : 0;
        for (unsigned c0 = 0; c0 < ncols; c0 += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c0 + k < ncols) s[k] = nxt[k];
            if (c0 + 8 < ncols) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + 8 + k < ncols) nxt[k] = lde[(size_t)(c0 + 8 + k) * col_stride + j];
            }
            poseidon::permute_residues(s);   // the capacity goes on as residues; only the digest is made canonical
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = gl::canon(s[k]);
    }
    ulonglong2* d = reinterpret_cast<ulonglong2*>(digests + 4 * j);
    d[0] = make_ulonglong2(s[0], s[1]);
    d[1] = make_ulonglong2(s[2], s[3]);
    if (sample) {
        clock_sample[0] = __builtin_amdgcn_s_memtime() - c0;
        clock_sample[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

__global__ void __launch_bounds__(THREADS)
fri_leaf_hash_kernel(const u64* __restrict__ v0, const u64* __restrict__ v1, size_t n_leaves, unsigned arity_bits,
                     u64* __restrict__ digests) {
    const size_t l = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (l >= n_leaves) return;
    const unsigned arity = 1u << arity_bits;
    const size_t base = l << arity_bits;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = 0;
    if (2 * arity <= 4) {
        for (unsigned m = 0; m < arity; ++m) { s[2 * m] = v0[base + m]; s[2 * m + 1] = v1[base + m]; }
    } else {
        // leaf = [v_0.c0, v_0.c1, v_1.c0, ...]; one absorb block = 4 extension values
        for (unsigned m0 = 0; m0 < arity; m0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (m0 + k < arity) { s[2 * k] = v0[base + m0 + k]; s[2 * k + 1] = v1[base + m0 + k]; }
            poseidon::permute_residues(s);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = gl::canon(s[k]);
    }
    ulonglong2* d = reinterpret_cast<ulonglong2*>(digests + 4 * l);
    d[0] = make_ulonglong2(s[0], s[1]);
    d[1] = make_ulonglong2(s[2], s[3]);
}

__global__ void __launch_bounds__(THREADS)
merkle_level_kernel(const u64* __restrict__ children, u64* __restrict__ parents, size_t n_parents) {
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (i >= n_parents) return;
    const ulonglong2* c = reinterpret_cast<const ulonglong2*>(children + 8 * i);
    const ulonglong2 a = c[0], b = c[1], e = c[2], f = c[3];
    u64 s[12] = {a.x, a.y, b.x, b.y, e.x, e.y, f.x, f.y, 0, 0, 0, 0};
    poseidon::permute_residues(s);
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = gl::canon(s[k]);   // the digest; the other eight words are dropped
    ulonglong2* d = reinterpret_cast<ulonglong2*>(parents + 4 * i);
    d[0] = make_ulonglong2(s[0], s[1]);
    d[1] = make_ulonglong2(s[2], s[3]);
}

// ---- 16-lanes-per-permutation variants for small node counts (latency-bound levels) ----
__global__ void __launch_bounds__(THREADS)
merkle_level_wide_kernel(const u64* __restrict__ children, u64* __restrict__ parents, size_t n_parents) {
    __shared__ u64 lds[(THREADS / poseidon::WIDE_LANES) * poseidon::WIDE_LDS_WORDS];
    const unsigned l = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t i = blockIdx.x * (size_t)(THREADS / 16) + g;
    const bool live = i < n_parents;
    u64 x = (live && l < 8) ? children[8 * i + l] : 0;
    x = poseidon::permute_wide(x, lds + g * poseidon::WIDE_LDS_WORDS, l);
    if (live && l < 4) parents[4 * i + l] = x;
}

// Several consecutive (small) levels in one launch: a workgroup owns 2^levels adjacent nodes and climbs to their common
// ancestor, keeping the intermediate digests in LDS (and writing every level to the tree, which is kept for the query
// openings).  The upper levels of a tree are a chain of dependent permutations; per-level launches cost a launch + drain
// (~6 us) on top of each ~10 us permutation, this form only the permutation.
constexpr unsigned CLIMB_THREADS = 512, CLIMB_MAX_LEVELS = 6;
struct ClimbArgs {
    const u64* children;          // 2^levels nodes per workgroup, consecutive
    u64* out[CLIMB_MAX_LEVELS];   // out[k]: level k + 1 above the children (n_children >> (k + 1) nodes)
    unsigned levels;
};
__global__ void __launch_bounds__(CLIMB_THREADS) merkle_climb_wide_kernel(ClimbArgs a) {
    constexpr unsigned G = CLIMB_THREADS / poseidon::WIDE_LANES;   // 32 permutations per pass
    __shared__ u64 lds[G * poseidon::WIDE_LDS_WORDS];
    __shared__ u64 buf[2][4 << (CLIMB_MAX_LEVELS - 1)];           // digests of the current / next level
    const unsigned l = threadIdx.x & 15, g = threadIdx.x >> 4;
    const unsigned in_per_wg = 1u << a.levels;
    const u64* src = a.children + (size_t)blockIdx.x * in_per_wg * 4;
    for (unsigned k = 0; k < a.levels; ++k) {
        const unsigned m = in_per_wg >> (k + 1);  // parents of this level inside the workgroup
        u64* dst = a.out[k] + (size_t)blockIdx.x * m * 4;
        for (unsigned i0 = 0; i0 < m; i0 += G) {
            // a wave (4 groups) whose groups all lie beyond m skips the pass as a whole: permute_wide needs the 64 lanes of a
            // wave together, not the workgroup, and the idle waves then cost no issue slots
            if (i0 + (g & ~3u) >= m) continue;
            const unsigned i = i0 + g;
            const bool live = i < m;
            u64 x = 0;
            if (live && l < 8) x = k == 0 ? src[8 * i + l] : buf[(k - 1) & 1][8 * i + l];
            x = poseidon::permute_wide(x, lds + g * poseidon::WIDE_LDS_WORDS, l);
            if (live && l < 4) {
                dst[4 * i + l] = x;
                buf[k & 1][4 * i + l] = x;
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(THREADS)
fri_leaf_hash_wide_kernel(const u64* __restrict__ v0, const u64* __restrict__ v1, size_t n_leaves, unsigned arity_bits,
                          u64* __restrict__ digests) {
    __shared__ u64 lds[(THREADS / poseidon::WIDE_LANES) * poseidon::WIDE_LDS_WORDS];
    const unsigned l = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t leaf = blockIdx.x * (size_t)(THREADS / 16) + g;
    const bool live = leaf < n_leaves;
    const unsigned n_elems = 2u << arity_bits;  // > 4 guaranteed by the caller
    const size_t base = leaf << arity_bits;
    u64 x = 0;
    for (unsigned e0 = 0; e0 < n_elems; e0 += 8) {
        const unsigned e = e0 + l;
        if (l < 8 && e < n_elems && live) x = ((e & 1) ? v1 : v0)[base + (e >> 1)];  // overwrite-mode absorb
        x = poseidon::permute_wide(x, lds + g * poseidon::WIDE_LDS_WORDS, l);
    }
    if (live && l < 4) digests[4 * leaf + l] = x;
}

__global__ void __launch_bounds__(THREADS) permute_batch_kernel(u64* states, size_t n) {
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = states[12 * i + k];
    poseidon::permute(s);
#pragma unroll
    for (int k = 0; k < 12; ++k) states[12 * i + k] = s[k];
}

__global__ void __launch_bounds__(THREADS)
hash_rows_kernel(const u64* __restrict__ rows, size_t n, unsigned len, u64* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x;
    if (i >= n) return;
    const u64* row = rows + i * len;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = 0;
    for (unsigned c0 = 0; c0 < len; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c0 + k < len) s[k] = row[c0 + k];
        poseidon::permute_residues(s);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * i + k] = gl::canon(s[k]);
}

struct PowState {
    u64 s[12];
};

// Candidates start .. start + count in ascending rounds of `stride`: thread i tries start + i, start + i + stride, ...  Before each permutation it
// looks at *result and leaves when a SMALLER valid nonce is already known (so the answer is still the smallest valid nonce of the range): with a
// 16-bit grind the expected answer is near 2^16 and the expected work that plus half a round, not the whole 2^17 range.  stride = count is the
// one-shot grid (launch_pow_search chooses).
__global__ void __launch_bounds__(THREADS)
pow_search_kernel(PowState st, unsigned pos, unsigned pow_bits, u64 start, u64 count, u64 stride, unsigned long long* result) {
    const u64 i = blockIdx.x * (u64)THREADS + threadIdx.x;
    if (i >= stride) return;
    for (u64 off = i; off < count; off += stride) {
        const u64 cand = start + off;
        if (__atomic_load_n(result, __ATOMIC_RELAXED) < cand) return;
        u64 s[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) s[k] = (unsigned)k == pos ? cand : st.s[k];
        poseidon::permute_residues(s);
        // pow_response = last element of squeeze() = state[7]; leading_zeros(response) >= pow_bits
        if (pow_bits == 0 || (gl::canon(s[7]) >> (64 - pow_bits)) == 0) {
            atomicMin(result, (unsigned long long)cand);
            return;   // this thread's later candidates are larger
        }
    }
}
}  // namespace

void launch_leaf_hash(hipStream_t s, const u64* lde, unsigned ncols, size_t n_leaves, size_t col_stride, u64* digests, u64* clock_sample) {
    if (clock_sample)
        hipLaunchKernelGGL(leaf_hash_kernel<true>, dim3((n_leaves + THREADS - 1) / THREADS), dim3(THREADS), 0, s, lde, ncols, n_leaves,
                           col_stride, digests, clock_sample);
    else
        hipLaunchKernelGGL(leaf_hash_kernel<false>, dim3((n_leaves + THREADS - 1) / THREADS), dim3(THREADS), 0, s, lde, ncols, n_leaves,
                           col_stride, digests, clock_sample);
}
// the defaults of a context's launch heuristics: below wide_threshold independent permutations a launch is latency-bound and the 16-lane
// form wins (measured)
Tuning Tuning::from_env() {
    Tuning t;
    if (const char* e = getenv("VPBS_WIDE_THRESHOLD")) t.wide_threshold = t.fri_leaf_wide_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("VPBS_FRI_LEAF_WIDE_THRESHOLD")) t.fri_leaf_wide_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("VPBS_MERKLE_CLIMB")) t.merkle_climb = atoi(e) != 0;
    if (const char* e = getenv("VPBS_GATES_FUSED")) t.gates_fused = atoi(e) != 0;
    if (const char* e = getenv("VPBS_GATE_ITEMS")) t.gate_items = (unsigned)std::max(1, atoi(e));
    if (const char* e = getenv("VPBS_GATES_TILE")) t.gates_tile = atoi(e) != 0;
    return t;
}

void launch_fri_leaf_hash(hipStream_t s, const Tuning& tune, const u64* v0, const u64* v1, size_t n_leaves, unsigned arity_bits, u64* digests) {
    if (n_leaves <= tune.fri_leaf_wide_threshold && (2u << arity_bits) > 4) {
        hipLaunchKernelGGL(fri_leaf_hash_wide_kernel, dim3((unsigned)((n_leaves * 16 + THREADS - 1) / THREADS)), dim3(THREADS), 0, s, v0,
                           v1, n_leaves, arity_bits, digests);
        return;
    }
    hipLaunchKernelGGL(fri_leaf_hash_kernel, dim3((n_leaves + THREADS - 1) / THREADS), dim3(THREADS), 0, s, v0, v1, n_leaves,
                       arity_bits, digests);
}
void launch_merkle_level(hipStream_t s, const Tuning& tune, const u64* children, u64* parents, size_t n_parents) {
    if (n_parents <= tune.wide_threshold) {
        hipLaunchKernelGGL(merkle_level_wide_kernel, dim3((unsigned)((n_parents * 16 + THREADS - 1) / THREADS)), dim3(THREADS), 0, s,
                           children, parents, n_parents);
        return;
    }
    hipLaunchKernelGGL(merkle_level_kernel, dim3((n_parents + THREADS - 1) / THREADS), dim3(THREADS), 0, s, children, parents,
                       n_parents);
}
void launch_merkle_tree(hipStream_t s, const Tuning& tune, u64* digests, const size_t* level_off, unsigned n_levels, size_t n_leaves) {
    unsigned k = 1;
    // big levels: one launch each (the chip is full)
    for (; k < n_levels && (n_leaves >> k) > tune.wide_threshold; ++k)
        launch_merkle_level(s, tune, digests + level_off[k - 1], digests + level_off[k], n_leaves >> k);
    // the latency-bound rest: up to CLIMB_MAX_LEVELS levels per launch
    const bool fused = tune.merkle_climb;
    while (k < n_levels) {
        const size_t n_children = n_leaves >> (k - 1);
        unsigned levels = std::min(CLIMB_MAX_LEVELS, n_levels - k);
        if (!fused || levels < 2) {
            launch_merkle_level(s, tune, digests + level_off[k - 1], digests + level_off[k], n_leaves >> k);
            ++k;
            continue;
        }
        ClimbArgs a{};
        a.children = digests + level_off[k - 1];
        a.levels = levels;
        for (unsigned j = 0; j < levels; ++j) a.out[j] = digests + level_off[k + j];
        hipLaunchKernelGGL(merkle_climb_wide_kernel, dim3((unsigned)(n_children >> levels)), dim3(CLIMB_THREADS), 0, s, a);
        k += levels;
    }
}
void launch_permute_batch(hipStream_t s, u64* states, size_t n) {
    hipLaunchKernelGGL(permute_batch_kernel, dim3((n + THREADS - 1) / THREADS), dim3(THREADS), 0, s, states, n);
}
void launch_hash_rows(hipStream_t s, const u64* rows, size_t n, unsigned len, u64* out) {
    hipLaunchKernelGGL(hash_rows_kernel, dim3((n + THREADS - 1) / THREADS), dim3(THREADS), 0, s, rows, n, len, out);
}
void launch_pow_search(hipStream_t s, const Tuning& tune, const u64* state12_host, unsigned pos, unsigned pow_bits, u64 start, u64 count,
                       u64* d_result) {
    PowState st;
    for (int k = 0; k < 12; ++k) st.s[k] = state12_host[k];
    // The same choice as the 16-lane Poseidon form (Tuning::wide_threshold): a context that has the GPU to itself hashes the whole range at
    // once (two waves per SIMD: 68 us for 2^17 candidates; rounds of 2^16 at one wave per SIMD take as long EACH, 94 us per proof on average,
    // rounds of 2^15 180 us); a context that shares the GPU with other chains (threshold lowered) runs rounds of 2^15 -- the others fill the
    // gaps and only the instruction count is left, which the early exits cut by a third.
    const bool shared_gpu = tune.wide_threshold < ((size_t)1 << 14);
    const u64 stride = shared_gpu ? std::min<u64>(count, (u64)1 << 15) : count;
    hipLaunchKernelGGL(pow_search_kernel, dim3((unsigned)((stride + THREADS - 1) / THREADS)), dim3(THREADS), 0, s, st, pos, pow_bits,
                       start, count, stride, reinterpret_cast<unsigned long long*>(d_result));
}
}  // namespace vpbs
