/* ORACLE (test infrastructure).  Eight Poseidon permutations at once, one per 64-bit lane of an AVX-512 register -- the checker's FAST
 * form of hash/poseidon.rs `Poseidon::poseidon`, written for the cpu_baseline leg of bench.py (a CPU figure measured with a permutation of
 * the class plonky2's own vectorised one is in, instead of the naive scalar one of poseidon.c) and for the Merkle trees of the oracle.
 *
 * It follows the NAIVE round structure of poseidon.c (30 rounds: constant layer, S-box x^7 on every element in the 8 full rounds and on
 * element 0 in the 22 partial rounds, dense MDS layer) -- no fused or "fast" partial rounds -- so that it is checked against orc_poseidon
 * round for round; the speed comes from the lanes and from lazy reductions (any u64 residue between rounds, MDS on 32-bit halves with
 * 64-bit accumulators and one fold per element).  tests/test_oracle_cpu.py compares it with the KAT-pinned scalar permutation on the
 * upstream vectors and on random states.  Compiled with per-function target attributes: the rest of the oracle stays x86-64-v2 and
 * orc_poseidon_x8_available() decides at run time (ORC_POSEIDON_X8=0 switches it off). */
#include "vpbs_oracle.h"
#include "poseidon_constants.h"
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define X8 __attribute__((target("avx512f,avx512dq"), always_inline)) static inline
#define X8_FN __attribute__((target("avx512f,avx512dq")))
typedef __m512i V;

static int x8_state = -1;
int orc_poseidon_x8_available(void) {
    if (x8_state < 0) {
        const char* e = getenv("ORC_POSEIDON_X8");
        x8_state = (!e || atoi(e) != 0) && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    }
    return x8_state;
}
/* tests and the cpu_baseline leg switch between the two forms inside one process; returns what is in force afterwards */
int orc_poseidon_x8_enable(int on) {
    x8_state = on && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    return x8_state;
}

X8 V bc(u64 x) { return _mm512_set1_epi64((long long)x); }
X8 V lo32(V a) { return _mm512_and_si512(a, bc(GL_EPS)); }
X8 V hi32(V a) { return _mm512_srli_epi64(a, 32); }

/* (lo + hi 2^64) mod p as some u64 residue: lo - hi_hi + hi_lo (2^32 - 1), each wrap corrected with 2^64 = 2^32 - 1 */
X8 V reduce128(V lo, V hi) {
    const V hh = hi32(hi), hl = lo32(hi);
    V t = _mm512_sub_epi64(lo, hh);
    t = _mm512_mask_sub_epi64(t, _mm512_cmplt_epu64_mask(lo, hh), t, bc(GL_EPS));
    const V u = _mm512_sub_epi64(_mm512_slli_epi64(hl, 32), hl);
    const V r = _mm512_add_epi64(t, u);
    return _mm512_mask_add_epi64(r, _mm512_cmplt_epu64_mask(r, u), r, bc(GL_EPS));
}
/* any residues in, a residue out: four 32 x 32 -> 64 products (vpmuludq) */
X8 V mul(V a, V b) {
    const V a1 = hi32(a), b1 = hi32(b);
    const V p00 = _mm512_mul_epu32(a, b), p10 = _mm512_mul_epu32(a1, b), p01 = _mm512_mul_epu32(a, b1), p11 = _mm512_mul_epu32(a1, b1);
    const V mid = _mm512_add_epi64(p10, hi32(p00));              /* < 2^64: (2^32-1)^2 + 2^32 - 1 */
    const V mid2 = _mm512_add_epi64(p01, lo32(mid));
    const V lo = _mm512_or_si512(_mm512_slli_epi64(mid2, 32), lo32(p00));
    const V hi = _mm512_add_epi64(_mm512_add_epi64(p11, hi32(mid)), hi32(mid2));
    return reduce128(lo, hi);
}
X8 V sbox7(V x) {
    const V x2 = mul(x, x), x4 = mul(x2, x2), x3 = mul(x2, x);
    return mul(x3, x4);
}
/* a + c for a canonical constant c: one wrap correction (the sum of a residue and a value below p wraps at most once) */
X8 V add_const(V a, u64 c) {
    const V cv = bc(c), s = _mm512_add_epi64(a, cv);
    return _mm512_mask_add_epi64(s, _mm512_cmplt_epu64_mask(s, cv), s, bc(GL_EPS));
}
X8 V canon(V x) { return _mm512_mask_sub_epi64(x, _mm512_cmpge_epu64_mask(x, bc(GL_P)), x, bc(GL_P)); }

/* st[i] = lane-wise state element i of eight permutations; canonical in, canonical out */
X8_FN void orc_poseidon_x8(V st[12]) {
    static const unsigned C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    for (int r = 0; r < 30; ++r) {
        V d[12];
        for (int i = 0; i < 12; ++i) d[i] = add_const(st[i], POSEIDON_RC[12 * r + i]);
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; ++i) d[i] = sbox7(d[i]);
        } else {
            d[0] = sbox7(d[0]);
        }
        /* MDS on the 32-bit halves: row sums are 272 (+ 8 on the diagonal), so the accumulators stay below 2^41 */
        V lo[12], hi[12];
        for (int i = 0; i < 12; ++i) { lo[i] = lo32(d[i]); hi[i] = hi32(d[i]); }
        for (int row = 0; row < 12; ++row) {
            V al = _mm512_setzero_si512(), ah = _mm512_setzero_si512();
            for (int i = 0; i < 12; ++i) {
                const V c = bc(C[i]);
                const int j = (i + row) % 12;
                al = _mm512_add_epi64(al, _mm512_mul_epu32(lo[j], c));
                ah = _mm512_add_epi64(ah, _mm512_mul_epu32(hi[j], c));
            }
            if (row == 0) {  /* MDS_MATRIX_DIAG[0] = 8 */
                al = _mm512_add_epi64(al, _mm512_slli_epi64(lo[0], 3));
                ah = _mm512_add_epi64(ah, _mm512_slli_epi64(hi[0], 3));
            }
            /* al + ah 2^32 = (al + (ah << 32) mod 2^64) + (ah >> 32 + carry) 2^64 */
            const V L = _mm512_add_epi64(al, _mm512_slli_epi64(ah, 32));
            const V H = _mm512_mask_add_epi64(hi32(ah), _mm512_cmplt_epu64_mask(L, al), hi32(ah), bc(1));
            st[row] = reduce128(L, H);
        }
    }
    for (int i = 0; i < 12; ++i) st[i] = canon(st[i]);
}

/* n independent states [n][12], in place */
X8_FN void orc_poseidon_batch_x8(u64* states, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i0 = 0; i0 < n; i0 += 8) {
        u64 buf[12][8] __attribute__((aligned(64)));
        const size_t cnt = n - i0 < 8 ? n - i0 : 8;
        for (size_t l = 0; l < 8; ++l)
            for (int k = 0; k < 12; ++k) buf[k][l] = states[12 * (i0 + (l < cnt ? l : 0)) + k];
        V st[12];
        for (int k = 0; k < 12; ++k) st[k] = _mm512_load_si512(buf[k]);
        orc_poseidon_x8(st);
        for (int k = 0; k < 12; ++k) _mm512_store_si512(buf[k], st[k]);
        for (size_t l = 0; l < cnt; ++l)
            for (int k = 0; k < 12; ++k) states[12 * (i0 + l) + k] = buf[k][l];
    }
}

/* hash_or_noop of `count` <= 8 rows of `len` elements each (row r at rows + r * stride): out[r][4].  Rows of at most four elements
 * are padded, not hashed; longer ones go through the overwrite-mode sponge, eight rows per permutation. */
X8_FN void orc_hash_rows_x8(const u64* rows, size_t stride, size_t len, size_t count, u64* out) {
    if (len <= 4) {
        for (size_t r = 0; r < count; ++r) orc_hash_or_noop(rows + r * stride, len, out + 4 * r);
        return;
    }
    u64 buf[12][8] __attribute__((aligned(64)));
    V st[12];
    for (int k = 0; k < 12; ++k) st[k] = _mm512_setzero_si512();
    for (size_t off = 0; off < len; off += 8) {
        const size_t blk = len - off < 8 ? len - off : 8;
        for (size_t k = 0; k < blk; ++k) {
            for (size_t l = 0; l < 8; ++l) buf[k][l] = rows[(l < count ? l : 0) * stride + off + k];
            st[k] = _mm512_load_si512(buf[k]);   /* overwrite mode */
        }
        orc_poseidon_x8(st);
    }
    for (int k = 0; k < 4; ++k) _mm512_store_si512(buf[k], st[k]);
    for (size_t r = 0; r < count; ++r)
        for (int k = 0; k < 4; ++k) out[4 * r + k] = buf[k][r];
}

/* two_to_one of `count` <= 8 consecutive pairs: children [2 count][4] -> parents [count][4] */
X8_FN void orc_two_to_one_x8(const u64* children, size_t count, u64* parents) {
    u64 buf[12][8] __attribute__((aligned(64)));
    V st[12];
    for (int k = 0; k < 8; ++k) {
        for (size_t l = 0; l < 8; ++l) buf[k][l] = children[8 * (l < count ? l : 0) + k];
        st[k] = _mm512_load_si512(buf[k]);
    }
    for (int k = 8; k < 12; ++k) st[k] = _mm512_setzero_si512();
    orc_poseidon_x8(st);
    for (int k = 0; k < 4; ++k) _mm512_store_si512(buf[k], st[k]);
    for (size_t r = 0; r < count; ++r)
        for (int k = 0; k < 4; ++k) parents[4 * r + k] = buf[k][r];
}

/* proof-of-work scan (fri/prover.rs fri_proof_of_work): the smallest w >= start for which the duplex of `state` with w written at `pos`
 * gives a response (state[7] after the permutation) with `pow_bits` leading zeros; eight candidates per permutation, in order */
X8_FN u64 orc_pow_search_x8(const u64 state[12], unsigned pos, unsigned pow_bits, u64 start) {
    u64 buf[8] __attribute__((aligned(64)));
    for (u64 base = start;; base += 8) {
        V st[12];
        for (int k = 0; k < 12; ++k) st[k] = bc(state[k]);
        for (int l = 0; l < 8; ++l) buf[l] = base + l;
        st[pos] = _mm512_load_si512(buf);
        orc_poseidon_x8(st);
        _mm512_store_si512(buf, st[7]);
        for (int l = 0; l < 8; ++l)
            if (pow_bits == 0 || (buf[l] >> (64 - pow_bits)) == 0) return base + l;
    }
}
#else
int orc_poseidon_x8_available(void) { return 0; }
int orc_poseidon_x8_enable(int on) { (void)on; return 0; }
void orc_poseidon_batch_x8(u64* states, size_t n) { (void)states; (void)n; }
void orc_hash_rows_x8(const u64* rows, size_t stride, size_t len, size_t count, u64* out) { (void)rows; (void)stride; (void)len; (void)count; (void)out; }
void orc_two_to_one_x8(const u64* children, size_t count, u64* parents) { (void)children; (void)count; (void)parents; }
u64 orc_pow_search_x8(const u64 state[12], unsigned pos, unsigned pow_bits, u64 start) { (void)state; (void)pos; (void)pow_bits; return start; }
#endif
