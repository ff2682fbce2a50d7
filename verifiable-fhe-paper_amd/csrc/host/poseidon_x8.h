// Eight independent Poseidon permutations side by side on the HOST: one permutation per 64-bit lane of AVX-512 registers.  The same
// schedule as poseidon::permute (poseidon.h: 4 full rounds, 7 fused groups of three partial rounds + one, 4 full rounds; the MDS layer on
// 32-bit halves with 64-bit accumulators and one 96 -> 64-bit fold per element), every scalar operation replaced by its 8-lane form:
// vpmuludq is exactly the 32 x 32 -> 64 multiply-add the scalar code is made of.  Used where the host has BATCHES of independent
// permutations -- the PoseidonGate rows of one dependency level of an in-circuit verifier's witness (the 28 FRI queries x 4 oracles:
// csrc/witness.hip), the Merkle paths of vpbs_verify_step -- not for chains (transcript, hash chains), where each permutation needs the
// one before.  Replaces nothing of plonky2 by itself: it is the host's form of hash/poseidon.rs `Poseidon::poseidon`.
// Compiled for the host only, with per-function target attributes (the rest of the library stays baseline x86-64); available() checks the CPU.
#pragma once
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>

#include <atomic>
#include <cstdlib>

#include "../poseidon.h"

namespace poseidon_x8 {
using gl::u32;
using gl::u64;
using V = __m512i;
#define X8 __attribute__((target("avx512f,avx512dq"), always_inline)) inline
#define X8_FN __attribute__((target("avx512f,avx512dq")))

inline bool available() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    return ok;
}
// vpbs_host_set_poseidon_x8: -1 = not set (the environment variable VPBS_POSEIDON_X8 decides, default on)
inline std::atomic<int>& switch_state() {
    static std::atomic<int> s{-1};
    return s;
}
inline bool enabled() {
    int s = switch_state().load(std::memory_order_relaxed);
    if (s < 0) {
        const char* e = std::getenv("VPBS_POSEIDON_X8");
        s = !(e && std::atoi(e) == 0);
        switch_state().store(s, std::memory_order_relaxed);
    }
    return s != 0 && available();
}

X8 V bc(u64 x) { return _mm512_set1_epi64((long long)x); }
X8 V add(V a, V b) { return _mm512_add_epi64(a, b); }
X8 V sub(V a, V b) { return _mm512_sub_epi64(a, b); }
X8 V mul32(V a, V b) { return _mm512_mul_epu32(a, b); }   // low 32 bits of every lane of a times those of b -> 64 bits
X8 V shr32(V a) { return _mm512_srli_epi64(a, 32); }
X8 V shl32(V a) { return _mm512_slli_epi64(a, 32); }
X8 V lo32(V a) { return _mm512_and_si512(a, bc(0xFFFFFFFFull)); }

// a + b (b canonical or not: any u64 residues), one wrap correction -- the scalar add_nc: s = a + b; if it wrapped, + eps
X8 V add_nc(V a, V b) {
    const V s = add(a, b);
    return _mm512_mask_add_epi64(s, _mm512_cmplt_epu64_mask(s, b), s, bc(gl::EPS));
}
// y - x for canonical y, x: canonical (gl::sub)
X8 V sub_canon(V y, V x) {
    const V d = sub(y, x);
    return _mm512_mask_add_epi64(d, _mm512_cmplt_epu64_mask(y, x), d, bc(gl::P));
}
X8 V canon(V x) { return _mm512_mask_sub_epi64(x, _mm512_cmpge_epu64_mask(x, bc(gl::P)), x, bc(gl::P)); }

// gl::mul_wide + gl::reduce128_nc, lane by lane: any u64 residues in, a u64 residue out
X8 V mul_nc(V a, V b) {
    const V a1 = shr32(a), b1 = shr32(b);
    const V t0 = mul32(a, b);
    const V t1 = add(mul32(a1, b), shr32(t0));
    const V t2 = add(mul32(a, b1), lo32(t1));
    const V lo = _mm512_or_si512(shl32(t2), lo32(t0));
    const V hi = add(add(mul32(a1, b1), shr32(t1)), shr32(t2));
    const V hi_hi = shr32(hi), hi_lo = lo32(hi);
    V t = sub(lo, hi_hi);
    t = _mm512_mask_sub_epi64(t, _mm512_cmplt_epu64_mask(lo, hi_hi), t, bc(gl::EPS));   // borrow: -2^64 = -eps
    const V u = sub(shl32(hi_lo), hi_lo);                                               // hi_lo (2^32 - 1)
    const V r = add(t, u);
    return _mm512_mask_add_epi64(r, _mm512_cmplt_epu64_mask(r, u), r, bc(gl::EPS));
}
X8 V sbox(V x) {
    const V x2 = mul_nc(x, x), x4 = mul_nc(x2, x2), x3 = mul_nc(x2, x);
    return mul_nc(x3, x4);
}
// acc_lo + acc_hi 2^32 (both < 2^58) -> a u64 residue (poseidon::fold96, host form)
X8 V fold96(V acc_lo, V acc_hi) {
    const V L = add(acc_lo, shl32(acc_hi));
    const V H = _mm512_mask_add_epi64(shr32(acc_hi), _mm512_cmplt_epu64_mask(L, acc_lo), shr32(acc_hi), bc(1));
    const V t1 = sub(shl32(H), H);
    const V v = add(L, t1);
    return _mm512_mask_add_epi64(v, _mm512_cmplt_epu64_mask(v, t1), v, bc(gl::EPS));
}
// s <- MDS s + k (k: 12 scalar constants or nullptr)
X8 void mds_add_const(V* s, const u64* kc) {
    static const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    V hi[12], c[12];
    for (int i = 0; i < 12; ++i) {
        hi[i] = shr32(s[i]);
        c[i] = bc(C[i]);
    }
    V out[12];
    for (int r = 0; r < 12; ++r) {
        const u64 k = kc ? kc[r] : 0;
        V al = bc((u32)k), ah = bc(k >> 32);
        for (int i = 0; i < 12; ++i) {
            al = add(al, mul32(s[(i + r) % 12], c[i]));
            ah = add(ah, mul32(hi[(i + r) % 12], c[i]));
        }
        if (r == 0) {   // MDS_MATRIX_DIAG[0] = 8
            al = add(al, mul32(s[0], bc(8)));
            ah = add(ah, mul32(hi[0], bc(8)));
        }
        out[r] = fold96(al, ah);
    }
    for (int r = 0; r < 12; ++r) s[r] = out[r];
}
// three partial rounds in one dense pass (poseidon::partial_group3_core, plain form); x_out: the S-box inputs of rounds 2 and 3 (canonical)
X8 void partial_group3(V* s, int g, V* x_out) {
    const poseidon::PartialGroup& G = poseidon::PG_HOST[g];
    s[0] = sbox(s[0]);
    V hi[12];
    for (int j = 0; j < 12; ++j) hi[j] = shr32(s[j]);
    V al = bc((u32)G.k2), ah = bc(G.k2 >> 32), bl = bc((u32)G.k3), bh = bc(G.k3 >> 32);
    for (int j = 0; j < 12; ++j) {
        const V m1 = bc(poseidon::MDS1[0][j]), m2 = bc(poseidon::MDS2[0][j]);
        al = add(al, mul32(s[j], m1));
        ah = add(ah, mul32(hi[j], m1));
        bl = add(bl, mul32(s[j], m2));
        bh = add(bh, mul32(hi[j], m2));
    }
    const V x2 = canon(fold96(al, ah));
    const V d2 = sub_canon(canon(sbox(x2)), x2);
    const V d2h = shr32(d2), m100 = bc(poseidon::MDS1[0][0]);
    bl = add(bl, mul32(d2, m100));
    bh = add(bh, mul32(d2h, m100));
    const V x3 = canon(fold96(bl, bh));
    const V d3 = sub_canon(canon(sbox(x3)), x3);
    const V d3h = shr32(d3);
    if (x_out) {
        x_out[0] = x2;
        x_out[1] = x3;
    }
    V out[12];
    for (int i = 0; i < 12; ++i) {
        V cl = bc((u32)G.kvec[i]), ch = bc(G.kvec[i] >> 32);
        for (int j = 0; j < 12; ++j) {
            const V m3 = bc(poseidon::MDS3[i][j]);
            cl = add(cl, mul32(s[j], m3));
            ch = add(ch, mul32(hi[j], m3));
        }
        const V m2 = bc(poseidon::MDS2[i][0]), m1 = bc(poseidon::MDS1[i][0]);
        cl = add(cl, add(mul32(d2, m2), mul32(d3, m1)));
        ch = add(ch, add(mul32(d2h, m2), mul32(d3h, m1)));
        out[i] = fold96(cl, ch);
    }
    for (int i = 0; i < 12; ++i) s[i] = out[i];
}

X8 void full_round(V* s, int round, V* rec) {
    for (int i = 0; i < 12; ++i) {
        if (rec) rec[i] = canon(s[i]);
        s[i] = sbox(s[i]);
    }
    mds_add_const(s, round + 1 < 30 ? poseidon::RC_HOST + 12 * (round + 1) : nullptr);
}

// The permutation of 8 states (s[i]: element i of every state, any u64 residues in; canonical out).  With `gate` != nullptr it also returns
// what a PoseidonGate row carries besides inputs and outputs -- the S-box inputs, canonical: gate[0..36) rounds 1..3 (12 each),
// gate[36..58) the 22 partial rounds, gate[58..106) rounds 26..29.
X8_FN inline void permute(V s[12], V* gate) {
    for (int i = 0; i < 12; ++i) s[i] = add_nc(s[i], bc(poseidon::RC_HOST[i]));
    for (int round = 0; round < 4; ++round) full_round(s, round, gate && round ? gate + 12 * (round - 1) : nullptr);
    for (int g = 0; g < 7; ++g) {
        V x[2];
        if (gate) gate[36 + 3 * g] = canon(s[0]);
        partial_group3(s, g, x);
        if (gate) {
            gate[36 + 3 * g + 1] = x[0];
            gate[36 + 3 * g + 2] = x[1];
        }
    }
    if (gate) gate[36 + 21] = canon(s[0]);
    s[0] = sbox(s[0]);
    mds_add_const(s, poseidon::RC_HOST + 12 * 26);
    for (int round = 26; round < 30; ++round) full_round(s, round, gate ? gate + 58 + 12 * (round - 26) : nullptr);
    for (int i = 0; i < 12; ++i) s[i] = canon(s[i]);
}

// n states of 12 words each, permuted in place: eight at a time, the remainder padded with copies (host-side batches)
X8_FN inline void permute_many(u64* states, size_t n) {
    alignas(64) u64 buf[12][8];
    for (size_t base = 0; base < n; base += 8) {
        const size_t cnt = n - base < 8 ? n - base : 8;
        for (int i = 0; i < 12; ++i)
            for (size_t l = 0; l < 8; ++l) buf[i][l] = states[12 * (base + (l < cnt ? l : 0)) + i];
        V s[12];
        for (int i = 0; i < 12; ++i) s[i] = _mm512_load_si512(buf[i]);
        permute(s, nullptr);
        for (int i = 0; i < 12; ++i) _mm512_store_si512(buf[i], s[i]);
        for (size_t l = 0; l < cnt; ++l)
            for (int i = 0; i < 12; ++i) states[12 * (base + l) + i] = buf[i][l];
    }
}
#undef X8
#undef X8_FN
}  // namespace poseidon_x8
#define VPBS_HAVE_POSEIDON_X8 1
#endif
