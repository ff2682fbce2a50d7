/* ORACLE (test infrastructure only -- never linked into or called by the product path).
 *
 * Goldilocks field GF(p), p = 2^64 - 2^32 + 1, and its quadratic extension GF(p^2) = F[X]/(X^2 - 7).
 * Restates plonky2_field 0.2.0 (crates.io, pinned by /root/reference/Cargo.lock:396-399; source NOT under
 * /root/reference): field/src/goldilocks_field.rs (reduce128, canonical form), field/src/extension/quadratic.rs
 * (W = 7), as summarised in SURVEY.md Appendix A.1.  The reference selects these types at
 * /root/reference/src/main.rs:33-35 (PoseidonGoldilocksConfig, D = 2).
 * All values are kept canonical (< p) at every function boundary.
 */
#ifndef VPBS_ORACLE_GL_H
#define VPBS_ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define GL_GENERATOR 7ULL                     /* MULTIPLICATIVE_GROUP_GENERATOR = coset_shift() */
#define GL_TWO_ADIC_GEN 1753635133440165772ULL /* POWER_OF_TWO_GENERATOR = 7^((p-1)/2^32) */
#define GL_TWO_ADICITY 32

static inline u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    if (s < a) s += GL_EPS; /* 2^64 = eps (mod p) */
    if (s >= GL_P) s -= GL_P;
    return s;
}
static inline u64 gl_sub(u64 a, u64 b) { return a >= b ? a - b : a - b + GL_P; }
static inline u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
static inline u64 gl_reduce128(u64 lo, u64 hi) {
    u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS; /* 2^96 = -1 */
    u64 t1 = hi_lo * GL_EPS;
    u64 r = t0 + t1;
    if (r < t1) r += GL_EPS;
    if (r >= GL_P) r -= GL_P;
    return r;
}
static inline u64 gl_mul(u64 a, u64 b) {
    u128 x = (u128)a * b;
    return gl_reduce128((u64)x, (u64)(x >> 64));
}
static inline u64 gl_sqr(u64 a) { return gl_mul(a, a); }
static inline u64 gl_exp(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = gl_mul(r, b); b = gl_sqr(b); e >>= 1; }
    return r;
}
static inline u64 gl_inv(u64 a) { return gl_exp(a, GL_P - 2); }
static inline u64 gl_from_u64(u64 x) { return x >= GL_P ? x - GL_P : x; }
/* primitive_root_of_unity(k) = POWER_OF_TWO_GENERATOR^(2^(32-k)) */
static inline u64 gl_root_of_unity(unsigned k) {
    u64 g = GL_TWO_ADIC_GEN;
    for (unsigned i = k; i < GL_TWO_ADICITY; ++i) g = gl_sqr(g);
    return g;
}

/* GF(p^2): (a0,a1)(b0,b1) = (a0 b0 + 7 a1 b1, a0 b1 + a1 b0) */
typedef struct { u64 c[2]; } ext2;
static inline ext2 ext_make(u64 a, u64 b) { ext2 r = {{a, b}}; return r; }
static inline ext2 ext_from_base(u64 a) { return ext_make(a, 0); }
static inline ext2 ext_add(ext2 a, ext2 b) { return ext_make(gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])); }
static inline ext2 ext_sub(ext2 a, ext2 b) { return ext_make(gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])); }
static inline ext2 ext_mul(ext2 a, ext2 b) {
    u64 c0 = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(7, gl_mul(a.c[1], b.c[1])));
    u64 c1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    return ext_make(c0, c1);
}
static inline ext2 ext_scalar_mul(ext2 a, u64 s) { return ext_make(gl_mul(a.c[0], s), gl_mul(a.c[1], s)); }
static inline ext2 ext_inv(ext2 a) {
    /* (a0 + a1 X)^-1 = (a0 - a1 X) / (a0^2 - 7 a1^2) */
    u64 norm = gl_sub(gl_sqr(a.c[0]), gl_mul(7, gl_sqr(a.c[1])));
    u64 ni = gl_inv(norm);
    return ext_make(gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni));
}
static inline ext2 ext_exp(ext2 b, u64 e) {
    ext2 r = ext_from_base(1);
    while (e) { if (e & 1) r = ext_mul(r, b); b = ext_mul(b, b); e >>= 1; }
    return r;
}
static inline int ext_eq(ext2 a, ext2 b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1]; }

static inline size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
#endif
