// Gate constraints of a plonky2 0.2.0 circuit, written once over a field type F:
//   F = u64 (base field)  : prover side, one LDE point per GPU thread (gates/ eval_unfiltered_base_*)
//   F = gl::Ext (GF(p^2)) : verifier side at zeta, host (gates/ eval_unfiltered)
// Each function pushes the gate's constraints IN plonky2's ORDER into a sink; the sink folds them with the powers of
// alpha (plonk/vanishing_poly.rs evaluate_gate_constraints* + reduce_with_powers_multi).  Wires that plonky2 reads as
// extension-field / ExtensionAlgebra elements (D = 2 consecutive wires) are Alg<F> = F[X]/(X^2 - 7).
// Path: compute_quotient_polys inside prove(), /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364 (SURVEY.md 8a
// row a13); verifier: cd.verify at :446.  Restated from the published crate -- parity unpinned (no golden circuit).
#pragma once
#include "gl.h"
#include "poseidon.h"
#include "../../include/vpbs_prover.h"

// Device: everything inlines into the per-gate kernel.  Host (verifier, one point): the generic code is kept out of line --
// force-inlining 30 Poseidon rounds of GF(p^2) arithmetic into one x86 function costs minutes of compile time for nothing.
#if defined(__HIP_DEVICE_COMPILE__)
#define GATES_FN __host__ __device__ __forceinline__
#else
#define GATES_FN __host__ __device__ __attribute__((noinline))
#endif

namespace gates {
using gl::Ext;
using gl::u32;
using gl::u64;

template <class F> struct Fld;
template <> struct Fld<u64> {
    static GL_HD u64 lift(u64 c) { return c; }
};
template <> struct Fld<Ext> {
    static GL_HD Ext lift(u64 c) { return gl::ext(c); }
};
// The evaluators' arithmetic.  Generic form (GF(p^2) at zeta in the verifier, any host use): the canonical gl:: operations.  Base field on the
// GPU: ANY u64 residues in and out (gl::add_a / sub_a / mul_nc / dot2_nc / mad_nc) -- wires and constants arrive canonical, everything computed
// from them is a residue, and every consumer takes residues: these operations themselves, the Poseidon layers of poseidon.h and the sink
// (its multiply-accumulate splits a u64 into halves whatever its value).  No product pays the 4-instruction canon that a canonical
// addition or subtraction downstream would need, and the additions cost 5 instructions instead of 6 (round 5).
template <class F> GL_HD F fadd(F a, F b) { return gl::add(a, b); }
template <class F> GL_HD F fsub(F a, F b) { return gl::sub(a, b); }
template <class F> GL_HD F fmul(F a, F b) { return gl::mul(a, b); }
template <class F> GL_HD F fdot2(F a, F b, F c, F d) { return gl::add(gl::mul(a, b), gl::mul(c, d)); }   // a b + c d
template <class F> GL_HD F fmad(F a, F b, F c) { return gl::add(gl::mul(a, b), c); }                       // a b + c
#if defined(__HIP_DEVICE_COMPILE__)
GL_HD u64 fadd(u64 a, u64 b) { return gl::add_a(a, b); }
GL_HD u64 fsub(u64 a, u64 b) { return gl::sub_a(a, b); }
GL_HD u64 fmul(u64 a, u64 b) { return gl::mul_nc(a, b); }
GL_HD u64 fdot2(u64 a, u64 b, u64 c, u64 d) { return gl::dot2_nc(a, b, c, d); }   // the two 128-bit products added before ONE reduction
GL_HD u64 fmad(u64 a, u64 b, u64 c) { return gl::mad_nc(a, b, c); }
#endif
// field element times a base-field constant.  On the GPU a constant below 2^32 (MDS entries, 7, bases, weights are NOT) takes
// two multiply-adds and one fold instead of a full modular multiplication.
GL_HD u64 mulc(u64 a, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_constant_p(c) && c < (1ull << 26))
        return poseidon::fold96((u64)(u32)a * (u32)c, (u64)(u32)(a >> 32) * (u32)c);   // any residue in, a residue out
    return gl::mul_nc(a, c);
#else
    return gl::mul(a, c);
#endif
}
GL_HD Ext mulc(Ext a, u64 c) { return gl::mul(a, c); }

// a + b X, X^2 = 7 over F: QuadraticExtension (F = base) or ExtensionAlgebra<F::Extension, 2> (F = GF(p^2))
template <class F> struct Alg {
    F a, b;
};
template <class F> GL_HD Alg<F> operator+(Alg<F> x, Alg<F> y) { return Alg<F>{fadd(x.a, y.a), fadd(x.b, y.b)}; }
template <class F> GL_HD Alg<F> operator-(Alg<F> x, Alg<F> y) { return Alg<F>{fsub(x.a, y.a), fsub(x.b, y.b)}; }
template <class F> GATES_FN Alg<F> operator*(Alg<F> x, Alg<F> y) {
    return Alg<F>{gl::add(gl::mul(x.a, y.a), mulc(gl::mul(x.b, y.b), 7)), gl::add(gl::mul(x.a, y.b), gl::mul(x.b, y.a))};
}
#if defined(__HIP_DEVICE_COMPILE__)
// Base field on the GPU: (a + b X)(c + d X) = (a c + 7 b d) + (a d + b c) X as two FUSED products -- the 128-bit products of a component
// are added before ONE reduction (gl::dot2_nc: 29 instructions instead of two multiplications and a modular addition, 48) -- and 7 d is
// two multiply-adds and a fold.  times7 / mul_lazy expose the pieces for loops that multiply by one fixed element (the ReducingGates'
// alpha); like every base-field operation of the GPU evaluators they take and return u64 residues.
GL_HD u64 times7(u64 x) { return poseidon::fold96((u64)(u32)x * 7u, (u64)(u32)(x >> 32) * 7u); }   // any residue in, a residue out
// x * y with y7 = times7(y.b)
GL_HD Alg<u64> mul_lazy(Alg<u64> x, Alg<u64> y, u64 y7) {
    return Alg<u64>{gl::dot2_nc(x.a, y.a, x.b, y7), gl::dot2_nc(x.a, y.b, x.b, y.a)};
}
GL_HD Alg<u64> operator*(Alg<u64> x, Alg<u64> y) {   // a plain overload: preferred to the template above for the base field
    return mul_lazy(x, y, times7(y.b));
}
// e t + v p (the step of the barycentric interpolation)
GL_HD Alg<u64> fma2(Alg<u64> e, Alg<u64> t, Alg<u64> v, Alg<u64> p) {
    const Alg<u64> x = mul_lazy(e, t, times7(t.b)), y = mul_lazy(v, p, times7(p.b));
    return Alg<u64>{gl::add_a(x.a, y.a), gl::add_a(x.b, y.b)};
}
// x + b (y - x) (RandomAccessGate's binary selection): one fused multiply-add
GL_HD u64 select_lerp(u64 x, u64 y, u64 b) { return gl::mad_nc(b, gl::sub_a(y, x), x); }
#endif
template <class F> GL_HD F times7(F x) { return mulc(x, 7); }
template <class F> GATES_FN Alg<F> mul_lazy(Alg<F> x, Alg<F> y, F) { return x * y; }
template <class F> GATES_FN Alg<F> fma2(Alg<F> e, Alg<F> t, Alg<F> v, Alg<F> p) { return e * t + v * p; }
template <class F> GL_HD F select_lerp(F x, F y, F b) { return gl::add(x, gl::mul(b, gl::sub(y, x))); }
template <class F> GL_HD Alg<F> scale(Alg<F> x, F s) { return Alg<F>{fmul(x.a, s), fmul(x.b, s)}; }  // scalar_mul
template <class F> GL_HD Alg<F> scalec(Alg<F> x, u64 c) { return Alg<F>{mulc(x.a, c), mulc(x.b, c)}; }
template <class F> GL_HD Alg<F> sub_base(Alg<F> x, u64 c) { return Alg<F>{fsub(x.a, Fld<F>::lift(c)), x.b}; }

// ---- Poseidon layers over F (hash/poseidon.rs constant_layer / sbox_layer / mds_layer and their *_field forms) ----
template <class F> struct Pos {
    static GATES_FN F sbox(F x) {
        const F x2 = gl::mul(x, x), x4 = gl::mul(x2, x2), x3 = gl::mul(x2, x);
        return gl::mul(x3, x4);
    }
    // s <- MDS s + round constants of `next_round` (none when next_round < 0)
    static GATES_FN void mds_then_constants(F* s, int next_round) {
        const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        F out[12];
        for (int r = 0; r < 12; ++r) {
            F acc = Fld<F>::lift(next_round >= 0 ? poseidon::rc(12 * next_round + r) : 0);
            for (int i = 0; i < 12; ++i) acc = gl::add(acc, mulc(s[(i + r) % 12], C[i]));
            if (r == 0) acc = gl::add(acc, mulc(s[0], 8));  // MDS_MATRIX_DIAG = [8, 0, ..]
            out[r] = acc;
        }
        for (int r = 0; r < 12; ++r) s[r] = out[r];
    }
};
#if defined(__HIP_DEVICE_COMPILE__)
// base field on the GPU: the multiply-add MDS and the hand-scheduled S-box of the hashing kernels (poseidon.h)
template <> struct Pos<u64> {
    static __device__ __forceinline__ u64 sbox(u64 x) { return poseidon::sbox(x); }
    static __device__ __forceinline__ void mds_then_constants(u64* s, int next_round) {
        if (next_round >= 0) {
            u64 kc[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) kc[i] = poseidon::rc(12 * next_round + i);
            poseidon::mds_add_const(s, kc);
        } else {
            poseidon::mds_add_const(s, nullptr);
        }
    }
};
#endif

// Vars concept:  using F; F wire(unsigned); F constant(unsigned) (selectors removed); u64 pi_hash(unsigned)
// Sink concept:  void push(F constraint)
template <class F, class V> GL_HD Alg<F> wire_alg(const V& v, unsigned start) { return Alg<F>{v.wire(start), v.wire(start + 1)}; }
template <class F, class S> GL_HD void push_alg(S& s, Alg<F> x) {
    s.push(x.a);
    s.push(x.b);
}

// gates/constant.rs: local_constants[i] - wire_output(i)
template <class F, class V, class S> GATES_FN void eval_constant(const vpbs_gate& g, const V& v, S& s) {
    for (unsigned i = 0; i < g.p0; ++i) s.push(fsub(v.constant(i), v.wire(i)));
}
// gates/public_input.rs: wires 0..4 - public_inputs_hash
template <class F, class V, class S> GATES_FN void eval_public_input(const vpbs_gate&, const V& v, S& s) {
    for (unsigned i = 0; i < 4; ++i) s.push(fsub(v.wire(i), Fld<F>::lift(v.pi_hash(i))));
}
// gates/arithmetic_base.rs: output - (m0 m1 c0 + addend c1), wires 4i .. 4i+3
template <class F, class V, class S> GATES_FN void eval_arithmetic(const vpbs_gate& g, const V& v, S& s) {
    const F c0 = v.constant(0), c1 = v.constant(1);
#pragma unroll 4
    for (unsigned i = 0; i < g.p0; ++i) {
        const F m0 = v.wire(4 * i), m1 = v.wire(4 * i + 1), addend = v.wire(4 * i + 2), out = v.wire(4 * i + 3);
        s.push(fsub(out, fdot2(fmul(m0, m1), c0, addend, c1)));
    }
}
// gates/base_sum.rs: reduce_with_powers(limbs, B) - sum; then prod_{k < B} (limb - k) per limb
// BASE: the base when it is known at compile time (2: every BaseSumGate of the reference's circuits -- the doubling and the single factor
// limb (limb - 1) then need no loop over the base and no multiplication by it), 0: g.p1 at run time
template <class F, unsigned BASE, class V, class S> GATES_FN void eval_base_sum_b(const vpbs_gate& g, const V& v, S& s) {
    const unsigned n = g.p0, B = BASE ? BASE : g.p1;
    // limbs are read in batches of 8 so that the loads of a batch are in flight together (the GPU thread is otherwise bound by
    // one memory latency per limb)
    F acc = Fld<F>::lift(0);
    for (unsigned hi = n; hi > 0;) {
        const unsigned lo = hi >= 8 ? hi - 8 : 0;
        F l[8];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u) l[u] = lo + u < hi ? v.wire(1 + lo + u) : Fld<F>::lift(0);
#pragma unroll
        for (unsigned u = 8; u-- > 0;)
            if (lo + u < hi) acc = fadd(B == 2 ? fadd(acc, acc) : mulc(acc, B), l[u]);
        hi = lo;
    }
    s.push(fsub(acc, v.wire(0)));
    for (unsigned i0 = 0; i0 < n; i0 += 8) {
        F l[8];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u) l[u] = i0 + u < n ? v.wire(1 + i0 + u) : Fld<F>::lift(0);
#pragma unroll
        for (unsigned u = 0; u < 8; ++u) {
            if (i0 + u >= n) break;
            F prod = l[u];
            for (unsigned k = 1; k < B; ++k) prod = fmul(prod, fsub(l[u], Fld<F>::lift(k)));
            s.push(prod);
        }
    }
}
template <class F, class V, class S> GATES_FN void eval_base_sum(const vpbs_gate& g, const V& v, S& s) {
    if (g.p1 == 2) eval_base_sum_b<F, 2>(g, v, s);
    else eval_base_sum_b<F, 0>(g, v, s);
}
// gates/poseidon.rs.  Wires: input 0..12, output 12..24, swap 24, delta 25..29, full_sbox_0(r=1..3) 29.., partial_sbox
// 65..87, full_sbox_1(r=0..3) 87..135.  The partial rounds are evaluated in the plain form (add constants, S-box on
// element 0, dense MDS): plonky2's "fast" partial rounds are a linear refactoring that leaves every S-box input -- and so
// every constraint -- unchanged.
template <class F, class V, class S> GATES_FN void eval_poseidon(const vpbs_gate&, const V& v, S& s) {
    const F swap = v.wire(24);
    s.push(fmul(swap, fsub(swap, Fld<F>::lift(1))));
    F st[12];
#pragma unroll
    for (unsigned i = 0; i < 4; ++i) {
        const F lhs = v.wire(i), rhs = v.wire(i + 4), delta = v.wire(25 + i);
        s.push(fsub(fmul(swap, fsub(rhs, lhs)), delta));
        st[i] = fadd(lhs, delta);
        st[i + 4] = fsub(rhs, delta);
    }
#pragma unroll
    for (unsigned i = 8; i < 12; ++i) st[i] = v.wire(i);
#pragma unroll
    for (unsigned i = 0; i < 12; ++i) st[i] = fadd(st[i], Fld<F>::lift(poseidon::rc(i)));  // constant_layer(round 0)
    int round = 0;
    for (unsigned r = 0; r < 4; ++r, ++round) {  // first full rounds
        if (r != 0) {
#pragma unroll
            for (unsigned i = 0; i < 12; ++i) {
                const F in = v.wire(29 + 12 * (r - 1) + i);
                s.push(fsub(st[i], in));
                st[i] = in;
            }
        }
#pragma unroll
        for (unsigned i = 0; i < 12; ++i) st[i] = Pos<F>::sbox(st[i]);
        Pos<F>::mds_then_constants(st, round + 1);
    }
    for (unsigned r = 0; r < 22; ++r, ++round) {  // partial rounds
        const F in = v.wire(65 + r);
        s.push(fsub(st[0], in));
        st[0] = Pos<F>::sbox(in);
        Pos<F>::mds_then_constants(st, round + 1);
    }
    for (unsigned r = 0; r < 4; ++r, ++round) {  // second full rounds
#pragma unroll
        for (unsigned i = 0; i < 12; ++i) {
            const F in = v.wire(87 + 12 * r + i);
            s.push(fsub(st[i], in));
            st[i] = in;
        }
#pragma unroll
        for (unsigned i = 0; i < 12; ++i) st[i] = Pos<F>::sbox(st[i]);
        Pos<F>::mds_then_constants(st, round + 1 < 30 ? round + 1 : -1);
    }
#pragma unroll
    for (unsigned i = 0; i < 12; ++i) s.push(fsub(v.wire(12 + i), st[i]));
}
// gates/poseidon_mds.rs: output_r - (MDS applied to 12 algebra elements); inputs 0..24, outputs 24..48
template <class F, class V, class S> GATES_FN void eval_poseidon_mds(const vpbs_gate&, const V& v, S& s) {
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    Alg<F> in[12];
#pragma unroll
    for (unsigned i = 0; i < 12; ++i) in[i] = wire_alg<F>(v, 2 * i);
#pragma unroll
    for (unsigned r = 0; r < 12; ++r) {
        Alg<F> acc{Fld<F>::lift(0), Fld<F>::lift(0)};
#pragma unroll
        for (unsigned i = 0; i < 12; ++i) acc = acc + scalec(in[(i + r) % 12], C[i]);
        if (r == 0) acc = acc + scalec(in[0], 8);
        push_alg(s, wire_alg<F>(v, 2 * (12 + r)) - acc);
    }
}
// gates/arithmetic_extension.rs: wires 8i: m0, m1, addend, output (2 each)
template <class F, class V, class S> GATES_FN void eval_arithmetic_ext(const vpbs_gate& g, const V& v, S& s) {
    const F c0 = v.constant(0), c1 = v.constant(1);
#pragma unroll 2
    for (unsigned i = 0; i < g.p0; ++i) {
        const Alg<F> m0 = wire_alg<F>(v, 8 * i), m1 = wire_alg<F>(v, 8 * i + 2), addend = wire_alg<F>(v, 8 * i + 4),
                     out = wire_alg<F>(v, 8 * i + 6);
        const Alg<F> mm = m0 * m1;
        push_alg(s, out - Alg<F>{fdot2(mm.a, c0, addend.a, c1), fdot2(mm.b, c0, addend.b, c1)});
    }
}
// gates/multiplication_extension.rs: wires 6i: m0, m1, output
template <class F, class V, class S> GATES_FN void eval_mul_ext(const vpbs_gate& g, const V& v, S& s) {
    const F c0 = v.constant(0);
#pragma unroll 2
    for (unsigned i = 0; i < g.p0; ++i) {
        const Alg<F> m0 = wire_alg<F>(v, 6 * i), m1 = wire_alg<F>(v, 6 * i + 2), out = wire_alg<F>(v, 6 * i + 4);
        push_alg(s, out - scale(m0 * m1, c0));
    }
}
// gates/reducing.rs: output 0..2, alpha 2..4, old_acc 4..6, coeffs 6..6+n (base), accs after; last acc = output
template <class F, class V, class S> GATES_FN void eval_reducing(const vpbs_gate& g, const V& v, S& s) {
    const unsigned n = g.p0;
    const Alg<F> alpha = wire_alg<F>(v, 2);
    const F alpha7 = times7(alpha.b);   // the one fixed multiplier of the loop
    Alg<F> acc = wire_alg<F>(v, 4);
#pragma unroll 4
    for (unsigned i = 0; i < n; ++i) {
        const Alg<F> next = wire_alg<F>(v, i == n - 1 ? 0 : 6 + n + 2 * i);
        Alg<F> c = mul_lazy(acc, alpha, alpha7);
        c.a = fadd(c.a, v.wire(6 + i));
        push_alg(s, c - next);
        acc = next;
    }
}
// gates/reducing_extension.rs: coeffs are extension elements (6 + 2i), accs after them
template <class F, class V, class S> GATES_FN void eval_reducing_ext(const vpbs_gate& g, const V& v, S& s) {
    const unsigned n = g.p0;
    const Alg<F> alpha = wire_alg<F>(v, 2);
    const F alpha7 = times7(alpha.b);
    Alg<F> acc = wire_alg<F>(v, 4);
#pragma unroll 4
    for (unsigned i = 0; i < n; ++i) {
        const Alg<F> next = wire_alg<F>(v, i == n - 1 ? 0 : 6 + 2 * n + 2 * i);
        push_alg(s, mul_lazy(acc, alpha, alpha7) + wire_alg<F>(v, 6 + 2 * i) - next);
        acc = next;
    }
}
// gates/random_access.rs: per copy: access_index, claimed_element, 2^bits list items; then extra constants; bit wires last.
// BITS is a template parameter so that the folded list stays in registers on the GPU.
template <class F, unsigned BITS, class V, class S> GATES_FN void eval_random_access_b(const vpbs_gate& g, const V& v, S& s) {
    constexpr unsigned vec = 1u << BITS;
    const unsigned copies = g.p1, extra = g.p2;
    const unsigned routed = (2 + vec) * copies + extra;
    for (unsigned c = 0; c < copies; ++c) {
        const unsigned base = (2 + vec) * c, bit0 = routed + c * BITS;
        F bit[BITS];
#pragma unroll
        for (unsigned b = 0; b < BITS; ++b) {
            bit[b] = v.wire(bit0 + b);
            s.push(fmul(bit[b], fsub(bit[b], Fld<F>::lift(1))));
        }
        F idx = Fld<F>::lift(0);
#pragma unroll
        for (unsigned b = BITS; b-- > 0;) idx = fadd(fadd(idx, idx), bit[b]);
        s.push(fsub(idx, v.wire(base)));
        F items[vec];
#pragma unroll
        for (unsigned i = 0; i < vec; ++i) items[i] = v.wire(base + 2 + i);
#pragma unroll
        for (unsigned b = 0; b < BITS; ++b) {
#pragma unroll
            for (unsigned i = 0; i < (vec >> (b + 1)); ++i)
                items[i] = select_lerp(items[2 * i], items[2 * i + 1], bit[b]);
        }
        s.push(fsub(items[0], v.wire(base + 1)));
    }
    for (unsigned i = 0; i < extra; ++i) s.push(fsub(v.constant(i), v.wire((2 + vec) * copies + i)));
}
template <class F, class V, class S> GATES_FN void eval_random_access(const vpbs_gate& g, const V& v, S& s) {
    switch (g.p0) {
        case 1: eval_random_access_b<F, 1>(g, v, s); break;
        case 2: eval_random_access_b<F, 2>(g, v, s); break;
        case 3: eval_random_access_b<F, 3>(g, v, s); break;
        case 4: eval_random_access_b<F, 4>(g, v, s); break;
        case 5: eval_random_access_b<F, 5>(g, v, s); break;
        default: break;  // rejected by vpbs_gates_layout
    }
}
// gates/exponentiation.rs: base 0, power bits 1..1+n (little endian), output 1+n, intermediate values 2+n..
template <class F, class V, class S> GATES_FN void eval_exponentiation(const vpbs_gate& g, const V& v, S& s) {
    const unsigned n = g.p0;
    const F base = v.wire(0), one = Fld<F>::lift(1);
    const F base_m1 = fsub(base, one);    // bit base + (1 - bit) = bit (base - 1) + 1
    F prev = one;
#pragma unroll 4
    for (unsigned i = 0; i < n; ++i) {
        const F sq = i == 0 ? one : fmul(prev, prev);
        const F bit = v.wire(1 + (n - 1 - i));
        const F computed = fmul(sq, fmad(bit, base_m1, one));
        const F cur = v.wire(2 + n + i);
        s.push(fsub(computed, cur));
        prev = cur;
    }
    s.push(fsub(v.wire(1 + n), prev));
}
// gates/coset_interpolation.rs.  Wires: shift 0, values 1..1+2*2^bits, evaluation_point, evaluation_value, then the
// intermediate (eval, prod) pairs and the shifted evaluation point.  domain / weights: two_adic_subgroup(bits) and its
// barycentric weights (computed by the caller, see coset_tables()).
struct CosetTables {
    u64 domain[32], weights[32];
};
template <class F, class V>
GL_HD void partial_interpolate(const CosetTables& t, const V& v, unsigned from, unsigned to, Alg<F> x, Alg<F>& eval, Alg<F>& prod) {
#pragma unroll 4
    for (unsigned i = from; i < to; ++i) {
        const Alg<F> val = scalec(wire_alg<F>(v, 1 + 2 * i), t.weights[i]);
        const Alg<F> term = sub_base(x, t.domain[i]);
        eval = fma2(eval, term, val, prod);
        prod = prod * term;
    }
}
template <class F, class V, class S> GATES_FN void eval_coset_interpolation(const vpbs_gate& g, const CosetTables& t, const V& v, S& s) {
    const unsigned bits = g.p0, degree = g.p1, points = 1u << bits;
    const unsigned n_inter = (points - 2) / (degree - 1);
    const unsigned start_point = 1 + 2 * points, start_value = start_point + 2, start_inter = start_value + 2;
    const unsigned start_shifted = start_inter + 4 * n_inter;
    const F shift = v.wire(0);
    const Alg<F> point = wire_alg<F>(v, start_point), shifted = wire_alg<F>(v, start_shifted);
    push_alg(s, point - scale(shifted, shift));
    Alg<F> eval{Fld<F>::lift(0), Fld<F>::lift(0)}, prod{Fld<F>::lift(1), Fld<F>::lift(0)};
    partial_interpolate<F>(t, v, 0, degree < points ? degree : points, shifted, eval, prod);
    for (unsigned i = 0; i < n_inter; ++i) {
        const Alg<F> ie = wire_alg<F>(v, start_inter + 2 * i), ip = wire_alg<F>(v, start_inter + 2 * (n_inter + i));
        push_alg(s, ie - eval);
        push_alg(s, ip - prod);
        const unsigned from = 1 + (degree - 1) * (i + 1);
        unsigned to = from + degree - 1;
        if (to > points) to = points;
        eval = ie;
        prod = ip;
        partial_interpolate<F>(t, v, from, to, shifted, eval, prod);
    }
    push_alg(s, wire_alg<F>(v, start_value) - eval);
}
inline CosetTables make_coset_tables(unsigned bits) {
    CosetTables t{};
    const unsigned n = 1u << bits;
    const u64 g = gl::root_of_unity(bits);
    u64 x = 1;
    for (unsigned i = 0; i < n; ++i, x = gl::mul(x, g)) t.domain[i] = x;
    for (unsigned i = 0; i < n; ++i) {  // barycentric_weights: 1 / prod_{j != i} (x_i - x_j)
        u64 d = 1;
        for (unsigned j = 0; j < n; ++j)
            if (j != i) d = gl::mul(d, gl::sub(t.domain[i], t.domain[j]));
        t.weights[i] = gl::inv(d);
    }
    return t;
}
// computed once per subgroup size (the inversions cost tens of microseconds of host time, and the prover asks every step)
inline const CosetTables& coset_tables(unsigned bits) {
    static const CosetTables tables[6] = {make_coset_tables(0), make_coset_tables(1), make_coset_tables(2),
                                          make_coset_tables(3), make_coset_tables(4), make_coset_tables(5)};
    return tables[bits <= 5 ? bits : 0];
}

// gates/gate.rs compute_filter: prod_{i in group, i != index} (i - s) [* (UNUSED_SELECTOR - s) with several selectors]
template <class F> GL_HD F compute_filter(const vpbs_gate& g, F sel, bool many_selectors) {
    F f = Fld<F>::lift(1);
    for (unsigned i = g.group_start; i < g.group_end; ++i)
        if (i != g.index) f = fmul(f, fsub(Fld<F>::lift(i), sel));
    if (many_selectors) f = fmul(f, fsub(Fld<F>::lift(VPBS_UNUSED_SELECTOR), sel));
    return f;
}

template <class F, class V, class S> GATES_FN void eval_gate(const vpbs_gate& g, const CosetTables* t, const V& v, S& s) {
    switch (g.kind) {
        case VPBS_GATE_NOOP: break;
        case VPBS_GATE_CONSTANT: eval_constant<F>(g, v, s); break;
        case VPBS_GATE_PUBLIC_INPUT: eval_public_input<F>(g, v, s); break;
        case VPBS_GATE_ARITHMETIC: eval_arithmetic<F>(g, v, s); break;
        case VPBS_GATE_BASE_SUM: eval_base_sum<F>(g, v, s); break;
        case VPBS_GATE_POSEIDON: eval_poseidon<F>(g, v, s); break;
        case VPBS_GATE_POSEIDON_MDS: eval_poseidon_mds<F>(g, v, s); break;
        case VPBS_GATE_ARITHMETIC_EXT: eval_arithmetic_ext<F>(g, v, s); break;
        case VPBS_GATE_MUL_EXT: eval_mul_ext<F>(g, v, s); break;
        case VPBS_GATE_REDUCING: eval_reducing<F>(g, v, s); break;
        case VPBS_GATE_REDUCING_EXT: eval_reducing_ext<F>(g, v, s); break;
        case VPBS_GATE_RANDOM_ACCESS: eval_random_access<F>(g, v, s); break;
        case VPBS_GATE_EXPONENTIATION: eval_exponentiation<F>(g, v, s); break;
        case VPBS_GATE_COSET_INTERPOLATION: eval_coset_interpolation<F>(g, *t, v, s); break;
        default: break;
    }
}
}  // namespace gates
