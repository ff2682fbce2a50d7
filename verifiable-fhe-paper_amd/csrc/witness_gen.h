// Witness generators of the supported gates and gadgets, written ONCE over a row accessor -- host rows, the slot rows of a compiled plan,
// the flag rows of plan creation, device rows: plonky2 0.2.0 gates/*.rs `SimpleGenerator::run_once` and iop/generator.rs -- and the
// copy-constraint partition (plonk/copy_constraint.rs, plonk/permutation_argument.rs Forest).  Included by witness.hip (host: one-shot
// generation, compiled plans, the two-phase runs of an IVC chain) and witness_device.hip (the same generators compiled for the device).
// Path: the first stage of prove(), /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "context.h"
#include "gates.h"
#include "host/poseidon_x8.h"
#include "../../include/vpbs_prover.h"

namespace vpbs {
namespace {
using gl::u32;
using gl::u64;
using A = gates::Alg<u64>;

struct GenError {
    std::string what;
};

// A gate instance owns `gen_count` generators (one per operation / copy, like plonky2); generator `sub` watches `deps` wires of
// its row and writes the wires it owns.
unsigned gen_count(const vpbs_gate& g) {
    switch (g.kind) {
        case VPBS_GATE_CONSTANT:
        case VPBS_GATE_ARITHMETIC:
        case VPBS_GATE_ARITHMETIC_EXT:
        case VPBS_GATE_MUL_EXT: return g.p0;
        case VPBS_GATE_RANDOM_ACCESS: return g.p1 + g.p2;  // one per copy + RandomAccessExtraConstants
        case VPBS_GATE_BASE_SUM:
        case VPBS_GATE_POSEIDON:
        case VPBS_GATE_POSEIDON_MDS:
        case VPBS_GATE_REDUCING:
        case VPBS_GATE_REDUCING_EXT:
        case VPBS_GATE_EXPONENTIATION:
        case VPBS_GATE_COSET_INTERPOLATION: return 1;
        default: return 0;  // NoopGate, PublicInputGate (its wires are copy-constrained to the in-circuit hash)
    }
}

void gen_deps(const vpbs_gate& g, unsigned sub, std::vector<unsigned>& d) {
    d.clear();
    auto range = [&](unsigned a, unsigned b) { for (unsigned i = a; i < b; ++i) d.push_back(i); };
    switch (g.kind) {
        case VPBS_GATE_ARITHMETIC: range(4 * sub, 4 * sub + 3); break;
        case VPBS_GATE_ARITHMETIC_EXT: range(8 * sub, 8 * sub + 6); break;
        case VPBS_GATE_MUL_EXT: range(6 * sub, 6 * sub + 4); break;
        case VPBS_GATE_BASE_SUM: d.push_back(0); break;
        case VPBS_GATE_POSEIDON: range(0, 12); d.push_back(24); break;
        case VPBS_GATE_POSEIDON_MDS: range(0, 24); break;
        case VPBS_GATE_REDUCING: range(2, 6 + g.p0); break;
        case VPBS_GATE_REDUCING_EXT: range(2, 6 + 2 * g.p0); break;
        case VPBS_GATE_RANDOM_ACCESS:
            if (sub < g.p1) {
                const unsigned vec = 1u << g.p0, base = (2 + vec) * sub;
                d.push_back(base);
                range(base + 2, base + 2 + vec);
            }
            break;
        case VPBS_GATE_EXPONENTIATION: range(0, 1 + g.p0); break;
        case VPBS_GATE_COSET_INTERPOLATION: range(0, 1 + 2 * (1u << g.p0) + 2); break;
        default: break;
    }
}

// R: get(wire) -> u64, set(wire, value)
template <class R> GL_HD A ralg(R& r, unsigned i) { return A{r.get(i), r.get(i + 1)}; }
template <class R> GL_HD void walg(R& r, unsigned i, A x) {
    r.set(i, x.a);
    r.set(i + 1, x.b);
}

// host and device (the device witness generator runs it once per PoseidonGate row and instance); failures go through the accessor
template <class R> GL_HD void poseidon_generate(R& r) {
    const u64 swap = r.get(24);
    if (swap > 1) {
        r.fail("PoseidonGate: swap wire is not boolean");
        return;
    }
    u64 st[12];
    for (int i = 0; i < 4; ++i) {
        const u64 lhs = r.get(i), rhs = r.get(i + 4);
        const u64 delta = gl::mul(swap, gl::sub(rhs, lhs));
        r.set(25 + i, delta);
        st[i] = gl::add(lhs, delta);
        st[i + 4] = gl::sub(rhs, delta);
    }
    for (int i = 8; i < 12; ++i) st[i] = r.get(i);
    // st holds the S-box inputs of the current round as arbitrary u64 residues; the MDS layer adds the next round's constants.  Same
    // schedule as poseidon::permute (the 22 partial rounds as 7 fused groups of three + one), with every S-box input the gate carries
    // as a wire written on the way.
    for (int i = 0; i < 12; ++i) st[i] = gl::add_nc(st[i], poseidon::rc(i));
    auto full_round = [&](int round, int wire0) {
        u64 kc[12];
        if (round + 1 < 30)
            for (int i = 0; i < 12; ++i) kc[i] = poseidon::rc(12 * (round + 1) + i);
        for (int i = 0; i < 12; ++i) {
            if (wire0 >= 0) r.set(wire0 + i, gl::canon(st[i]));
            st[i] = poseidon::sbox(st[i]);
        }
        poseidon::mds_add_const(st, round + 1 < 30 ? kc : nullptr);
    };
    for (int round = 0; round < 4; ++round) full_round(round, round ? 29 + 12 * (round - 1) : -1);
    for (int g = 0; g < 7; ++g) {
        u64 x[2];
        r.set(65 + 3 * g, gl::canon(st[0]));
        poseidon::partial_group3_core<false>(st, g, nullptr, x);
        r.set(65 + 3 * g + 1, x[0]);
        r.set(65 + 3 * g + 2, x[1]);
    }
    {
        u64 kc[12];
        for (int i = 0; i < 12; ++i) kc[i] = poseidon::rc(12 * 26 + i);
        r.set(65 + 21, gl::canon(st[0]));
        st[0] = poseidon::sbox(st[0]);
        poseidon::mds_add_const(st, kc);
    }
    for (int round = 26; round < 30; ++round) full_round(round, 87 + 12 * (round - 26));
    for (int i = 0; i < 12; ++i) r.set(12 + i, gl::canon(st[i]));
}

// host and device; `t`: the interpolation tables of a CosetInterpolationGate (gates::coset_tables(g.p0)), unused otherwise
template <class R> GL_HD void gen_run(const vpbs_gate& g, unsigned sub, const u64* c, R& r, const gates::CosetTables* t = nullptr) {
    switch (g.kind) {
        case VPBS_GATE_CONSTANT: r.set(sub, c[sub]); break;
        case VPBS_GATE_ARITHMETIC:
            r.set(4 * sub + 3, gl::add(gl::mul(gl::mul(r.get(4 * sub), r.get(4 * sub + 1)), c[0]), gl::mul(r.get(4 * sub + 2), c[1])));
            break;
        case VPBS_GATE_BASE_SUM: {  // BaseSplitGenerator: little-endian base-B digits of the canonical sum
            u64 x = r.get(0);
            if (g.p1 == 2) {
                for (unsigned i = 0; i < g.p0; ++i, x >>= 1) r.set(1 + i, x & 1);
            } else {
                for (unsigned i = 0; i < g.p0; ++i) {
                    r.set(1 + i, x % g.p1);
                    x /= g.p1;
                }
            }
            if (x != 0) r.fail("BaseSumGate: integer too large to fit in the given number of limbs");
            break;
        }
        case VPBS_GATE_POSEIDON: poseidon_generate(r); break;
        case VPBS_GATE_POSEIDON_MDS: {
            const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
            A in[12];
            for (unsigned i = 0; i < 12; ++i) in[i] = ralg(r, 2 * i);
            for (unsigned row = 0; row < 12; ++row) {
                A acc{0, 0};
                for (unsigned i = 0; i < 12; ++i) acc = acc + gates::scalec(in[(i + row) % 12], C[i]);
                if (row == 0) acc = acc + gates::scalec(in[0], 8);
                walg(r, 2 * (12 + row), acc);
            }
            break;
        }
        case VPBS_GATE_ARITHMETIC_EXT:
            walg(r, 8 * sub + 6, gates::scale(ralg(r, 8 * sub) * ralg(r, 8 * sub + 2), c[0]) + gates::scale(ralg(r, 8 * sub + 4), c[1]));
            break;
        case VPBS_GATE_MUL_EXT: walg(r, 6 * sub + 4, gates::scale(ralg(r, 6 * sub) * ralg(r, 6 * sub + 2), c[0])); break;
        case VPBS_GATE_REDUCING:
        case VPBS_GATE_REDUCING_EXT: {
            const bool ext = g.kind == VPBS_GATE_REDUCING_EXT;
            const unsigned n = g.p0, start_accs = ext ? 6 + 2 * n : 6 + n;
            const A alpha = ralg(r, 2);
            A acc = ralg(r, 4);
            for (unsigned i = 0; i < n; ++i) {
                const A coeff = ext ? ralg(r, 6 + 2 * i) : A{r.get(6 + i), 0};
                acc = acc * alpha + coeff;
                walg(r, i == n - 1 ? 0 : start_accs + 2 * i, acc);
            }
            break;
        }
        case VPBS_GATE_RANDOM_ACCESS: {
            const unsigned bits = g.p0, vec = 1u << bits, routed = (2 + vec) * g.p1 + g.p2;
            if (sub >= g.p1) {  // RandomAccessExtraConstantsGenerator
                const unsigned i = sub - g.p1;
                r.set((2 + vec) * g.p1 + i, c[i]);
                break;
            }
            const unsigned base = (2 + vec) * sub;
            const u64 idx = r.get(base);
            if (idx >= vec) {
                r.fail("RandomAccessGate: access index out of range");
                break;
            }
            r.set(base + 1, r.get(base + 2 + (unsigned)idx));
            for (unsigned b = 0; b < bits; ++b) r.set(routed + sub * bits + b, (idx >> b) & 1);
            break;
        }
        case VPBS_GATE_EXPONENTIATION: {
            const unsigned n = g.p0;
            const u64 base = r.get(0);
            u64 prev = 1;
            for (unsigned i = 0; i < n; ++i) {
                const u64 sq = i == 0 ? 1 : gl::mul(prev, prev);
                const u64 bit = r.get(1 + (n - 1 - i));
                if (bit > 1) {
                    r.fail("ExponentiationGate: power bit is not boolean");
                    return;
                }
                prev = bit ? gl::mul(sq, base) : sq;
                r.set(2 + n + i, prev);
            }
            r.set(1 + n, prev);
            break;
        }
        case VPBS_GATE_COSET_INTERPOLATION: {
            const unsigned points = 1u << g.p0, degree = g.p1, ni = (points - 2) / (degree - 1);
            const unsigned start_point = 1 + 2 * points, start_value = start_point + 2, start_inter = start_value + 2;
            const unsigned start_shifted = start_inter + 4 * ni;
            const u64 shift = r.get(0);
            if (shift == 0 || !t) {
                r.fail("CosetInterpolationGate: zero shift");
                break;
            }
            const A shifted = gates::scale(ralg(r, start_point), gl::inv(shift));
            walg(r, start_shifted, shifted);
            A eval{0, 0}, prod{1, 0};
            auto run = [&](unsigned from, unsigned to) {
                for (unsigned i = from; i < to; ++i) {
                    const A term = gates::sub_base(shifted, t->domain[i]);
                    eval = eval * term + gates::scalec(ralg(r, 1 + 2 * i), t->weights[i]) * prod;
                    prod = prod * term;
                }
            };
            run(0, degree < points ? degree : points);
            for (unsigned i = 0; i < ni; ++i) {
                walg(r, start_inter + 2 * i, eval);
                walg(r, start_inter + 2 * (ni + i), prod);
                const unsigned from = 1 + (degree - 1) * (i + 1);
                run(from, from + degree - 1 < points ? from + degree - 1 : points);
            }
            walg(r, start_value, eval);
            break;
        }
        default: break;
    }
}

inline const gates::CosetTables* tables_of(const vpbs_gate& g) {
    return g.kind == VPBS_GATE_COSET_INTERPOLATION ? &gates::coset_tables(g.p0) : nullptr;
}

struct PlainRow {
    u64* w;
    u64 get(unsigned i) const { return w[i]; }
    void set(unsigned i, u64 v) { w[i] = v; }
    void fail(const char* m) { throw GenError{m}; }
};

// ---- the copy-constraint partition (plonk/copy_constraint.rs + permutation_argument.rs Forest) ----
struct Partition {
    std::vector<u32> parent;
    explicit Partition(size_t n) : parent(n) { std::iota(parent.begin(), parent.end(), 0u); }
    u32 find(u32 x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    }
    void merge(u32 a, u32 b) {
        a = find(a);
        b = find(b);
        if (a != b) parent[a] = b;
    }
};

bool check_circuit(const vpbs_circuit* c) {
    if (!c || !c->gates || !c->n_gates || !c->row_gate || c->log_n == 0 || c->log_n > 24 || c->n_routed > c->n_wires) return false;
    if (c->n_copies && !c->copies) return false;
    const size_t n = (size_t)1 << c->log_n;
    for (size_t r = 0; r < n; ++r)
        if (c->row_gate[r] >= c->n_gates) return false;
    for (size_t i = 0; i < 2 * c->n_copies; ++i)
        if (c->copies[i] >= (size_t)c->n_routed * n) return false;
    for (unsigned i = 0; i < c->n_gates; ++i)
        if (c->gates[i].num_wires > c->n_wires) return false;
    if (c->n_generators && !c->generators) return false;
    const size_t total = (size_t)c->n_wires * n;
    for (size_t i = 0; i < c->n_generators; ++i) {
        const vpbs_generator& g = c->generators[i];
        if ((g.n_in && !g.in) || (g.n_out && !g.out)) return false;
        for (unsigned k = 0; k < g.n_in; ++k)
            if (g.in[k] >= total) return false;
        for (unsigned k = 0; k < g.n_out; ++k)
            if (g.out[k] >= total) return false;
        if (g.kind == VPBS_GEN_EQUALITY && (g.n_in != 2 || g.n_out != 2)) return false;
        if (g.kind == VPBS_GEN_BASE_SUM && (g.p0 < 2 || g.n_out != 1)) return false;
        if (g.kind == VPBS_GEN_WIRE_SPLIT && (g.p0 < 1 || g.p0 > 63 || g.n_in != 1 || g.n_out < 1)) return false;
        if (g.kind == VPBS_GEN_QUOTIENT_EXT && (g.n_in != 4 || g.n_out != 2)) return false;
        if (g.kind == VPBS_GEN_COPY && (g.n_in != 1 || g.n_out != 1)) return false;
        if (g.kind == VPBS_GEN_LOW_HIGH && (g.p0 < 1 || g.p0 > 63 || g.n_in != 1 || g.n_out != 2)) return false;
        if (g.kind > VPBS_GEN_LOW_HIGH) return false;
    }
    return true;
}
// ---- compiled witness generation -------------------------------------------------------------------------------------------------
// Which generator can run when depends only on the circuit and on WHICH targets the PartialWitness sets, never on the values: the
// readiness loop of generate_partial_witness is therefore run once, at plan creation, and recorded as a straight-line schedule over
// value slots (one slot per copy-constraint class).  A run is then: presets -> slots, the schedule in order (no readiness checks, no
// union-find), slots -> wires (full_witness) in parallel.  The step circuit is proven n + 2 times per PBS with the same plan.
}  // namespace
}  // namespace vpbs
