"""CPU restatement of the native TFHE data path of one vPBS step (TEST INFRASTRUCTURE).

What it follows in the reference (/root/reference/src/vtfhe/):
  * step semantics            ivc_based_vpbs.rs:99-125  (first step: rotate by -mask; middle: CMUX; last: external product only)
  * rotate_poly / mod switch  mod.rs:80-107 (round-to-nearest of the top log2(2N) bits, built from power-of-two rotations),
                              glwe_poly.rs:132-148 (`rotate`: multiplication by X^shift, negacyclic), crypto/lwe.rs:28-34
  * signed decomposition      glwe_poly.rs:28-50 (`decompose`), :150-166
  * GLEV / GGSW products      glev_ct.rs:92-110 (`mul`: top ELL limbs, NTT, vec_inner with the GLEV rows),
                              ggsw_ct.rs:98-112 (`external_product`: glev_muls[K-1] - sum of the others, inverse NTT)
  * plain crypto for tests    crypto/glwe.rs:49-63 (encrypt/decrypt), crypto/glev.rs:26-38, crypto/ggsw.rs:26-36
The negacyclic NTT itself is the C oracle's (pinned by TESTG/TESTGHAT).  Pure Python big-int arithmetic otherwise.
"""
import numpy as np

import oracle as orc

P = 0xFFFFFFFF00000001


def num_limbs(logb):
    return -(-64 // logb)


def mod_switch(mask, log_n_ring):
    """rotation amount in [0, 2N]: top log2(2N) bits of the canonical value, rounded with the next bit (mod.rs:85-106)"""
    log2n = log_n_ring + 1
    x = int(mask)
    return (x >> (64 - log2n)) + ((x >> (64 - log2n - 1)) & 1)


def rotate(poly, shift):
    """multiplication by X^shift modulo X^N + 1, 0 <= shift <= 2N (composition of GlwePoly::rotate steps)"""
    n = len(poly)
    out = [0] * n
    for i, c in enumerate(poly):
        j = i + shift
        sign = (j // n) & 1
        out[j % n] = (P - int(c)) % P if sign else int(c)
    return out


def decompose(x, logb):
    """glwe_poly.rs:28-50: centred base-2^logb digits (little-endian) as field elements"""
    nl = num_limbs(logb)
    tb = nl * logb
    x = int(x)
    sgn = (x >> (tb - 1)) & 1 if tb <= 64 else 0
    xc = (P - x) % P if sgn else x
    out, carry = [], 0
    for l in range(nl):
        k = (xc >> (l * logb)) & ((1 << logb) - 1)
        kw = k + carry
        carry = (k >> (logb - 1)) & 1
        bal = (kw - (carry << logb)) % P
        out.append((P - bal) % P if sgn else bal)
    return out


class Ring:
    def __init__(self, log_n):
        self.log_n, self.n = log_n, 1 << log_n
        self.roots, self.invroots, self.ninv = orc.negacyclic_params(log_n)

    def fw(self, poly):
        return [int(v) for v in orc.negacyclic_forward(np.array(poly, dtype=np.uint64), self.roots)]

    def bw(self, poly):
        return [int(v) for v in orc.negacyclic_backward(np.array(poly, dtype=np.uint64), self.invroots, self.ninv)]

    def mul(self, a, b):
        return self.bw([x * y % P for x, y in zip(self.fw(a), self.fw(b))])


def external_product(ring, ggsw_hat, glwe, K, ELL, logb):
    """ggsw_hat[p][l][r] = NTT-domain polynomial r of GLWE l of GLEV p; glwe: K coefficient polynomials"""
    nl = num_limbs(logb)
    muls = []
    for p in range(K):
        digits = [decompose(c, logb) for c in glwe[p]]
        limbs_hat = [ring.fw([d[nl - ELL + l] for d in digits]) for l in range(ELL)]
        muls.append([[sum(limbs_hat[l][i] * ggsw_hat[p][l][r][i] for l in range(ELL)) % P for i in range(ring.n)] for r in range(K)])
    out = []
    for r in range(K):
        acc = [(muls[K - 1][r][i] - sum(muls[p][r][i] for p in range(K - 1))) % P for i in range(ring.n)]
        out.append(ring.bw(acc))
    return out


def step(ring, acc_in, mask, ggsw_hat, K, ELL, logb, first_step=False, last_step=False):
    """ivc_based_vpbs.rs:99-125"""
    m = (P - int(mask)) % P if first_step else int(mask)
    s = mod_switch(m, ring.log_n)
    shifted = [rotate(p, s) for p in acc_in]
    if first_step:
        return shifted
    diff = [[(a - b) % P for a, b in zip(sp, ap)] for sp, ap in zip(shifted, acc_in)]
    xin = [list(map(int, p)) for p in acc_in] if last_step else diff
    xout = external_product(ring, ggsw_hat, xin, K, ELL, logb)
    if last_step:
        return xout
    return [[(a + int(b)) % P for a, b in zip(xp, ap)] for xp, ap in zip(xout, acc_in)]


# ---- noise-free plain TFHE for the property tests (crypto/*.rs with sigma = 0) ----
def glwe_encrypt(ring, rng, s, m, K):
    mask = [[int(v) for v in rng.integers(0, P, size=ring.n, dtype=np.uint64)] for _ in range(K - 1)]
    body = [0] * ring.n
    for a, sk in zip(mask, s):
        body = [(x + y) % P for x, y in zip(body, ring.mul(a, sk))]
    return mask + [[(b + int(mi)) % P for b, mi in zip(body, m)]]


def glwe_decrypt(ring, s, ct, K):
    mask = [0] * ring.n
    for a, sk in zip(ct[:K - 1], s):
        mask = [(x + y) % P for x, y in zip(mask, ring.mul(a, sk))]
    return [(b - x) % P for b, x in zip(ct[K - 1], mask)]


def ggsw_encrypt_hat(ring, rng, s, m, K, ELL, logb):
    """Ggsw::encrypt(...).ntt_forward(): GLEV p encrypts m * s_p (p < K-1) or m (p = K-1), gadget B^(first_limb + l)"""
    first = num_limbs(logb) - ELL
    out = []
    for p in range(K):
        mp = ring.mul(m, s[p]) if p < K - 1 else list(m)
        glev = []
        for l in range(ELL):
            scale = pow(2, logb * (first + l), P)
            ct = glwe_encrypt(ring, rng, s, [x * scale % P for x in mp], K)
            glev.append([ring.fw(poly) for poly in ct])
        out.append(glev)
    return out


def flatten_ggsw(ggsw_hat):
    return np.array([c for glev in ggsw_hat for glwe in glev for poly in glwe for c in poly], dtype=np.uint64)


# ---- the PBS around the accumulator chain (src/main.rs:40-65, crypto/mod.rs:17-45, crypto/lwe.rs, crypto/ggsw.rs:38-48) ----
def get_delta(two_p):
    return P >> (two_p - 1).bit_length()          # F::order() >> log2_ceil(2p)


def get_testv(ring, p, delta):
    block = ring.n // p
    coeffs = [i * delta % P for i in range(p) for _ in range(block)]
    s = block // 2                                  # Poly::left_shift(block / 2): c[i] <- c[i+s], wrapped terms negated
    return [coeffs[i + s] if i < ring.n - s else (P - coeffs[i - ring.n + s]) % P for i in range(ring.n)]


def pbs_setup(ring, rng, n, K, ELL, logb, p=2):
    """noise-free keys as main.rs builds them: partial key s_to (first n coefficients binary), LWE key = those
    coefficients, GLWE key, bootstrapping key (NTT domain), key-switching key"""
    s_to = [[int(v) for v in rng.integers(0, 2, size=n)] + [0] * (ring.n - n)] + [[0] * ring.n for _ in range(K - 1)]
    s_lwe = s_to[0][:n]
    s_glwe = [[int(v) for v in rng.integers(0, 2, size=ring.n)] for _ in range(K - 1)]
    bsk = [ggsw_encrypt_hat(ring, rng, s_glwe, [si] + [0] * (ring.n - 1), K, ELL, logb) for si in s_lwe]
    first = num_limbs(logb) - ELL
    ksk = []
    for i in range(K):                              # compute_ksk: GLEV i encrypts s_from[i] (i < K-1) or 1, under s_to
        msg = s_glwe[i] if i < K - 1 else [1] + [0] * (ring.n - 1)
        glev = []
        for l in range(ELL):
            scale = pow(2, logb * (first + l), P)
            ct = glwe_encrypt(ring, rng, s_to, [x * scale % P for x in msg], K)
            glev.append([ring.fw(poly) for poly in ct])
        ksk.append(glev)
    return s_to, s_lwe, s_glwe, bsk, ksk


def lwe_encrypt(rng, s, m):
    mask = [int(v) for v in rng.integers(0, P, size=len(s), dtype=np.uint64)]
    return mask + [(sum(a * b for a, b in zip(mask, s)) + m) % P]


def pbs_chain(ring, acc_init, ct, bsk, ksk, K, ELL, logb):
    n = len(ct) - 1
    accs = [step(ring, acc_init, ct[n], None, K, ELL, logb, first_step=True)]
    for x in range(n):
        accs.append(step(ring, accs[-1], ct[x], bsk[x], K, ELL, logb))
    accs.append(step(ring, accs[-1], 0, ksk, K, ELL, logb, last_step=True))
    return accs


# ---- seeded keys / ciphertexts (the checker of vpbs_keygen, vpbs_lwe_encrypt, vpbs_testv; generator spec: csrc/keygen.hip header) ----
# The encryptions follow the reference literally -- Glwe::encrypt in the coefficient domain (crypto/glwe.rs:49-57: mask, error,
# body = <s, mask> + error + m through Poly::mul), Glev / Ggsw::encrypt (crypto/glev.rs:26-38, crypto/ggsw.rs:26-36), then ntt_forward --
# while the product generates the NTT-domain form directly; both must give the same field elements.
_G = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1
S_TO, S_GLWE, BSK_MASK, BSK_NOISE, KSK_MASK, KSK_NOISE, LWE_MASK, LWE_NOISE = range(1, 9)


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


class Seeded:
    def __init__(self, seed):
        self.seed = seed

    def stream(self, kind, a=0, b=0, c=0):
        tag = (kind << 56) | (a << 32) | (b << 16) | c
        return _mix64((self.seed + _G * (tag + 1)) & _M64)

    @staticmethod
    def draw(key, i):
        return _mix64((key + _G * (i + 1)) & _M64)

    def field(self, key, i):
        u = self.draw(key, i)
        return u - P if u >= P else u

    def bit(self, key, i):
        return self.draw(key, i) & 1

    def noise(self, key, i, m_sigma):
        s = 0
        for k in range(6):
            d = self.draw(key, 6 * i + k)
            s += (d & 0xFFFFFFFF) + (d >> 32)
        t = s - (6 << 32)
        return ((t * m_sigma) >> 32) % P          # Python's >> floors, % maps negatives into the field


def sigma_to_int(sigma):
    import math
    return int(math.floor(sigma * float(P) + 0.5))


def seeded_keys(ring, seed, n_lwe, K):
    """partial_key / flatten_partial_key / key_gen (crypto/glwe.rs:15-40)"""
    g = Seeded(seed)
    s_to = [[0] * ring.n for _ in range(K)]
    for x in range(n_lwe):
        s_to[x // ring.n][x % ring.n] = g.bit(g.stream(S_TO, x // ring.n), x % ring.n)
    s_lwe = [s_to[x // ring.n][x % ring.n] for x in range(n_lwe)]
    s_glwe = [[g.bit(g.stream(S_GLWE, j), i) for i in range(ring.n)] for j in range(K - 1)]
    return s_to, s_lwe, s_glwe


def seeded_glwe_encrypt(ring, g, s, m, K, kind_mask, kind_noise, gi, pl, m_sigma):
    """Glwe::encrypt with the mask / error streams of GLWE (gi, pl)"""
    mask = [[g.field(g.stream(kind_mask, gi, pl, r), i) for i in range(ring.n)] for r in range(K - 1)]
    ke = g.stream(kind_noise, gi, pl, 0)
    body = [g.noise(ke, i, m_sigma) for i in range(ring.n)]
    for a, sk in zip(mask, s):
        body = [(x + y) % P for x, y in zip(body, ring.mul(a, sk))]
    return mask + [[(b + int(mi)) % P for b, mi in zip(body, m)]]


def seeded_ggsw_hat(ring, g, s, glev_msgs, bit, K, ELL, logb, kind_mask, kind_noise, gi, m_sigma):
    """Ggsw::encrypt(s, .).ntt_forward() where GLEV p encrypts bit * glev_msgs[p]; flattened in Ggsw::flatten order"""
    first = num_limbs(logb) - ELL
    out = []
    for p in range(K):
        for l in range(ELL):
            scale = pow(2, logb * (first + l), P) * bit
            ct = seeded_glwe_encrypt(ring, g, s, [x * scale % P for x in glev_msgs[p]], K, kind_mask, kind_noise, gi, p * ELL + l, m_sigma)
            out += [c for poly in ct for c in ring.fw(poly)]
    return np.array(out, dtype=np.uint64)


def seeded_pbs_keys(ring, seed, n_lwe, K, ELL, logb, sigma_glwe=0.0, sigma_lwe=0.0, bsk_indices=None):
    """main.rs:40-46 with seeded RNGs -> s_to, s_lwe, s_glwe, {i: bsk[i]} for the requested indices (all when None), ksk"""
    g = Seeded(seed)
    s_to, s_lwe, s_glwe = seeded_keys(ring, seed, n_lwe, K)
    one = [1] + [0] * (ring.n - 1)
    # compute_bsk: Ggsw::encrypt(s_glwe, constant(s_i)): GLEV p < K-1 encrypts s_i * s_glwe[p], the last one s_i
    msgs = [list(s_glwe[p]) for p in range(K - 1)] + [one]
    idx = range(n_lwe) if bsk_indices is None else bsk_indices
    bsk = {i: seeded_ggsw_hat(ring, g, s_glwe, msgs, s_lwe[i], K, ELL, logb, BSK_MASK, BSK_NOISE, i, sigma_to_int(sigma_glwe)) for i in idx}
    # compute_ksk(s_to, s_from = s_glwe): GLEV i < K-1 encrypts s_glwe[i], the last one the constant 1, under s_to
    ksk = seeded_ggsw_hat(ring, g, s_to, msgs, 1, K, ELL, logb, KSK_MASK, KSK_NOISE, 0, sigma_to_int(sigma_lwe))
    return s_to, s_lwe, s_glwe, bsk, ksk


def seeded_lwe_encrypt(seed, s_lwe, message, sigma_lwe, nonce=0):
    """lwe::encrypt (crypto/lwe.rs:55-64)"""
    g = Seeded(seed)
    km, ke = g.stream(LWE_MASK, nonce), g.stream(LWE_NOISE, nonce)
    mask = [g.field(km, i) for i in range(len(s_lwe))]
    body = (sum(a * b for a, b in zip(mask, s_lwe)) + message + g.noise(ke, 0, sigma_to_int(sigma_lwe))) % P
    return mask + [body]
