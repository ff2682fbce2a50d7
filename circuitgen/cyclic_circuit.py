"""TEST INFRASTRUCTURE (stand-in for the Rust circuit builder): the reference's CYCLIC step circuit -- build_step_circuit plus the recursive
verifier of its own previous proof -- as a circuit description for the product.

What it follows in the reference: /root/reference/src/vtfhe/ivc_based_vpbs.rs:159-275 (`verified_pbs`: the step circuit, the verifier data as
public inputs :214, the inner proof's public inputs unpacked and wired to the step :217-266, `conditionally_verify_cyclic_proof_or_dummy`
:268-274), :277-371 (base proof, the n + 2 chained proofs), :388-489 (`verify_pbs`).  What plonky2 0.2.0 does inside those calls is restated
from the published crate: recursion/recursive_verifier.rs `verify_proof` (public-input hash, `get_challenges`, `eval_vanishing_poly_circuit`
with every gate's `eval_unfiltered_circuit`, quotient recombination, `verify_fri_proof`), fri/recursive_verifier.rs (proof of work,
`fri_verify_initial_proof`, `fri_combine_initial`, `compute_evaluation` through the CosetInterpolationGate, Merkle paths with `permute_swapped`
and a RandomAccessGate into the cap), iop/challenger.rs `RecursiveChallenger`, recursion/cyclic_recursion.rs (verifier data in the public
inputs, inner == outer check, base case).  The verifier gadget mirrors, step by step, the product's host verifier (csrc/verifier.hip), which
verifies the same proofs natively.

Differences from plonky2, stated: row placement and gadget packing are this builder's own.  The base case is plonky2's: TWO proof slots (the
cyclic proof; the dummy circuit's proof with its public inputs and verifier data, the targets plonky2's DummyProofGenerator fills), every
word selected by `condition` (select_proof_with_pis), one verifier on the selection.  The circuit digest is CircuitBuilder::build's
(`circuit_digest` below: cap, hash_pad of the empty domain separator, degree bits), carried in the verifier data.  Parity with the crate is
unpinned; the semantics are those of the reference: ONE proof attests to the whole chain.
"""
import numpy as np

import pymodel
import step_circuit as sc
from step_circuit import ARITH_OPS, BASE_SUM_LIMBS, NUM_CONSTS, N_ROUTED, P, Builder

GATE_SPEC = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing",
             "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"]
AEXT_OPS, MEXT_OPS, RED_N, REDX_N = 10, 13, 43, 32          # *_from_config under standard_recursion_config (vpbs_gate_default_params)
RA_BITS, RA_VEC, RA_COPIES = 4, 16, 4
CI_BITS, CI_POINTS = 4, 16
CI_START_POINT, CI_START_VALUE = 1 + 2 * CI_POINTS, 3 + 2 * CI_POINTS
W7 = 7                                                      # GF(p^2) = GF(p)[X] / (X^2 - 7)


def inv(x):
    return pow(x % P, P - 2, P)


class ExtBuilder(Builder):
    """Builder + the gadgets of plonky2's recursive verifier over the 14 standard gates.  An extension target is a pair of targets."""
    gate_spec = GATE_SPEC

    def __init__(self):
        super().__init__()
        self.aext_free, self.mext_free, self.ra_free = {}, {}, None

    # ---- extension-field targets (gadgets/arithmetic_extension.rs) ----
    def cext(self, c0, c1=0):
        return (self.constant(c0), self.constant(c1))

    def zero_ext(self):
        return self.cext(0)

    def one_ext(self):
        return self.cext(1)

    def to_ext(self, t):
        return (t, self.zero())

    def virtual_ext(self):
        return (self.virtual(), self.virtual())

    def connect_ext(self, a, b):
        self.connect(a[0], b[0])
        self.connect(a[1], b[1])

    def aext(self, c0, a, b, c1, c):
        """c0 * a * b + c1 * c on an ArithmeticExtensionGate slot (wires 8 i: multiplicand 0, multiplicand 1, addend, output)"""
        key = (c0 % P, c1 % P)
        slot = self.aext_free.get(key)
        if slot is None:
            slot = (self.add_row("arithmetic_ext", list(key)), 0)
        row, i = slot
        if i + 1 < AEXT_OPS:
            self.aext_free[key] = (row, i + 1)
        else:
            self.aext_free.pop(key, None)
        for j, src in enumerate((a, b, c)):
            self._wire(row, 8 * i + 2 * j, src[0])
            self._wire(row, 8 * i + 2 * j + 1, src[1])
        return (self._wire(row, 8 * i + 6), self._wire(row, 8 * i + 7))

    def mul_ext(self, a, b):
        """a * b on a MulExtensionGate slot (wires 6 i: multiplicand 0, multiplicand 1, output)"""
        slot = self.mext_free.get(1)
        if slot is None:
            slot = (self.add_row("mul_ext", [1]), 0)
        row, i = slot
        if i + 1 < MEXT_OPS:
            self.mext_free[1] = (row, i + 1)
        else:
            self.mext_free.pop(1, None)
        for j, src in enumerate((a, b)):
            self._wire(row, 6 * i + 2 * j, src[0])
            self._wire(row, 6 * i + 2 * j + 1, src[1])
        return (self._wire(row, 6 * i + 4), self._wire(row, 6 * i + 5))

    def add_ext(self, a, b):
        return self.aext(1, self.one_ext(), a, 1, b)

    def sub_ext(self, a, b):
        return self.aext(1, self.one_ext(), a, P - 1, b)

    def mul_add_ext(self, a, b, c):
        return self.aext(1, a, b, 1, c)

    def mul_sub_ext(self, a, b, c):
        return self.aext(1, a, b, P - 1, c)

    def mulc_ext(self, a, c):
        """a times a base-field constant"""
        c %= P
        if c == 1:
            return a
        return self.mul_ext(a, self.cext(c))

    def addc_ext(self, a, c):
        return self.add_ext(a, self.cext(c))

    def scalar_mul_ext(self, s, a):
        """base-field target times extension target"""
        return self.mul_ext(self.to_ext(s), a)

    def square_ext(self, a):
        return self.mul_ext(a, a)

    def exp_pow2_ext(self, a, k):
        for _ in range(k):
            a = self.square_ext(a)
        return a

    def div_ext(self, num, den):
        """QuotientGeneratorExtension + the constraint quotient * den == num"""
        q = self.virtual_ext()
        self.generators.append(("quotient_ext", 0, [num[0], num[1], den[0], den[1]], [q[0], q[1]]))
        self.connect_ext(self.mul_ext(q, den), num)
        return q

    def select_ext(self, b, x, y):
        """gadgets/select.rs select_ext: b x - (b y - y) on two ArithmeticExtensionGate operations, b lifted to the extension"""
        be = self.to_ext(b)
        return self.mul_sub_ext(be, x, self.mul_sub_ext(be, y, y))

    # ---- util/reducing.rs ReducingFactorTarget ----
    def reduce_base(self, coeffs, alpha):
        """sum_j coeffs[j] alpha^j for base-field targets (ReducingGate: old_acc * alpha^n + Horner over its 43 coefficient wires)"""
        rev = list(reversed(coeffs))
        rev = [self.zero()] * (-len(rev) % RED_N) + rev          # leading zeros leave a zero accumulator zero
        acc = self.zero_ext()
        for at in range(0, len(rev), RED_N):
            row = self.add_row("reducing")
            self._wire(row, 2, alpha[0]); self._wire(row, 3, alpha[1])
            self._wire(row, 4, acc[0]); self._wire(row, 5, acc[1])
            for i, c in enumerate(rev[at:at + RED_N]):
                self._wire(row, 6 + i, c)
            acc = (self._wire(row, 0), self._wire(row, 1))
        return acc

    def reduce_ext(self, coeffs, alpha):
        """sum_j coeffs[j] alpha^j for extension targets (ReducingExtensionGate, 32 coefficients per row)"""
        rev = list(reversed(coeffs))
        rev = [self.zero_ext()] * (-len(rev) % REDX_N) + rev
        acc = self.zero_ext()
        for at in range(0, len(rev), REDX_N):
            row = self.add_row("reducing_ext")
            self._wire(row, 2, alpha[0]); self._wire(row, 3, alpha[1])
            self._wire(row, 4, acc[0]); self._wire(row, 5, acc[1])
            for i, c in enumerate(rev[at:at + REDX_N]):
                self._wire(row, 6 + 2 * i, c[0]); self._wire(row, 7 + 2 * i, c[1])
            acc = (self._wire(row, 0), self._wire(row, 1))
        return acc

    # ---- gadgets/random_access.rs ----
    def random_access(self, index, items):
        """items[index] for a 16-entry list (RandomAccessGate, bits = 4: copy c owns wires 18 c: index, claimed element, the list)"""
        assert len(items) == RA_VEC
        if self.ra_free is None:
            self.ra_free = (self.add_row("random_access", [0, 0]), 0)
        row, c = self.ra_free
        self.ra_free = (row, c + 1) if c + 1 < RA_COPIES else None
        base = (2 + RA_VEC) * c
        self._wire(row, base, index)
        for i, t in enumerate(items):
            self._wire(row, base + 2 + i, t)
        return self._wire(row, base + 1)

    def random_access_many(self, index, lists):
        """[l[index] for l in lists] (random_access_hash / random_access_extension)"""
        return [self.random_access(index, l) for l in lists]

    # ---- gadgets/interpolation.rs interpolate_coset ----
    def interpolate_coset(self, shift, values, point):
        """the polynomial through (shift * g^i, values[i]), i < 16, evaluated at `point` (CosetInterpolationGate, subgroup_bits = 4)"""
        assert len(values) == CI_POINTS
        row = self.add_row("coset_interpolation")
        self._wire(row, 0, shift)
        for i, v in enumerate(values):
            self._wire(row, 1 + 2 * i, v[0]); self._wire(row, 2 + 2 * i, v[1])
        self._wire(row, CI_START_POINT, point[0]); self._wire(row, CI_START_POINT + 1, point[1])
        return (self._wire(row, CI_START_VALUE), self._wire(row, CI_START_VALUE + 1))

    # ---- hash/poseidon.rs mds_layer_circuit through the PoseidonMdsGate ----
    def poseidon_mds(self, state):
        row = self.add_row("poseidon_mds")
        for i, s in enumerate(state):
            self._wire(row, 2 * i, s[0]); self._wire(row, 2 * i + 1, s[1])
        return [(self._wire(row, 24 + 2 * i), self._wire(row, 25 + 2 * i)) for i in range(12)]

    # ---- hash: one PoseidonGate row with its swap wire (gadgets/hash.rs permute_swapped) ----
    def permute(self, inputs, swap=None):
        row = self.add_row("poseidon")
        for i in range(12):
            self._wire(row, i, inputs[i])
        self._wire(row, 24, self.zero() if swap is None else swap)
        return [self._wire(row, 12 + i) for i in range(12)]

    def hash_or_noop(self, elems):
        if len(elems) <= 4:
            return list(elems) + [self.zero()] * (4 - len(elems))
        return self.hash_no_pad(elems)

    def exp_from_bits_const_base(self, base, bits):
        """base^(sum bits[i] 2^i) for a constant base: product *= 1 + bit (base^(2^i) - 1)   (gadgets/arithmetic.rs)"""
        product = self.one()
        for i, bit in enumerate(bits):
            product = self.arithmetic(pow(base, 1 << i, P) - 1, product, bit, 1, product)
        return product

    def exp_pow2(self, x, k):
        for _ in range(k):
            x = self.mul(x, x)
        return x

    def range_check(self, x, bits):
        """x < 2^bits: the base-2 limbs of a BaseSumGate sum to x (gadgets/range_check.rs)"""
        return self.split_le(x, bits)

    def fill_unused_slots(self):
        zero, ze = self.zero(), None
        for (c0, c1), (row, i) in list(self.aext_free.items()):
            for j in range(i, AEXT_OPS):
                for w in range(6):
                    self._wire(row, 8 * j + w, zero)
        self.aext_free.clear()
        for _, (row, i) in list(self.mext_free.items()):
            for j in range(i, MEXT_OPS):
                for w in range(4):
                    self._wire(row, 6 * j + w, zero)
        self.mext_free.clear()
        if self.ra_free is not None:
            row, c = self.ra_free
            for k in range(c, RA_COPIES):
                base = (2 + RA_VEC) * k
                self._wire(row, base, zero)
                for i in range(RA_VEC):
                    self._wire(row, base + 2 + i, zero)
            self.ra_free = None
        del ze


# ================= gate constraints at one extension point, written once over a backend =================
# gates/*.rs `eval_unfiltered` (numbers) / `eval_unfiltered_circuit` (targets), in the constraint order of csrc/gates.h (= plonky2's).
# Backend: lift(c) -> element, add, sub, mul, mulc(x, c), mds(list of 12 elements) -> MDS layer.  Elements are GF(p^2) values or
# extension targets.  Alg elements (plonky2's ExtensionAlgebra, wires read as pairs) are tuples (a, b) of backend elements, X^2 = 7.
class NumBackend:
    """GF(p^2) numbers as (c0, c1): the model the circuit backend is checked against (and vpbs_gate_terms_at is checked against)"""

    def lift(self, c):
        return (c % P, 0)

    def add(self, x, y):
        return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)

    def sub(self, x, y):
        return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)

    def mul(self, x, y):
        return pymodel.ext_mul(x, y)

    def mulc(self, x, c):
        return (x[0] * c % P, x[1] * c % P)

    def mds(self, s):
        C = pymodel.CIRC
        out = []
        for r in range(12):
            acc = (0, 0)
            for i in range(12):
                acc = self.add(acc, self.mulc(s[(i + r) % 12], C[i]))
            if r == 0:
                acc = self.add(acc, self.mulc(s[0], 8))
            out.append(acc)
        return out


class CircuitBackend:
    def __init__(self, cb):
        self.cb = cb

    def lift(self, c):
        return self.cb.cext(c)

    def add(self, x, y):
        return self.cb.add_ext(x, y)

    def sub(self, x, y):
        return self.cb.sub_ext(x, y)

    def mul(self, x, y):
        return self.cb.mul_ext(x, y)

    def mulc(self, x, c):
        return self.cb.mulc_ext(x, c)

    def mds(self, s):
        return self.cb.poseidon_mds(s)


def _alg_ops(B):
    def aadd(x, y):
        return (B.add(x[0], y[0]), B.add(x[1], y[1]))

    def asub(x, y):
        return (B.sub(x[0], y[0]), B.sub(x[1], y[1]))

    def amul(x, y):
        return (B.add(B.mul(x[0], y[0]), B.mulc(B.mul(x[1], y[1]), W7)), B.add(B.mul(x[0], y[1]), B.mul(x[1], y[0])))

    def ascale(x, s):
        return (B.mul(x[0], s), B.mul(x[1], s))

    def ascalec(x, c):
        return (B.mulc(x[0], c), B.mulc(x[1], c))
    return aadd, asub, amul, ascale, ascalec


def barycentric(bits):
    n = 1 << bits
    g = pymodel.root_of_unity(bits)
    dom = [pow(g, i, P) for i in range(n)]
    wts = []
    for i in range(n):
        d = 1
        for j in range(n):
            if j != i:
                d = d * (dom[i] - dom[j]) % P
        wts.append(inv(d))
    return dom, wts


def gate_constraints(B, kind, g, wire, const, pi_hash):
    """constraints of one gate, in order.  wire(i) / const(i): backend elements (const: gate constants, selectors removed); pi_hash(i):
    backend element of the i-th word of the public-input hash."""
    aadd, asub, amul, ascale, ascalec = _alg_ops(B)
    walg = lambda i: (wire(i), wire(i + 1))
    out = []
    push = out.append

    def push_alg(x):
        out.append(x[0])
        out.append(x[1])
    one = B.lift(1)
    if kind == "constant":
        for i in range(g.p0):
            push(B.sub(const(i), wire(i)))
    elif kind == "public_input":
        for i in range(4):
            push(B.sub(wire(i), pi_hash(i)))
    elif kind == "arithmetic":
        c0, c1 = const(0), const(1)
        for i in range(g.p0):
            m0, m1, addend, o = wire(4 * i), wire(4 * i + 1), wire(4 * i + 2), wire(4 * i + 3)
            push(B.sub(o, B.add(B.mul(B.mul(m0, m1), c0), B.mul(addend, c1))))
    elif kind == "base_sum":
        n, base = g.p0, g.p1
        acc = B.lift(0)
        for i in reversed(range(n)):
            acc = B.add(B.mulc(acc, base), wire(1 + i))
        push(B.sub(acc, wire(0)))
        for i in range(n):
            l = wire(1 + i)
            prod = l
            for k in range(1, base):
                prod = B.mul(prod, B.sub(l, B.lift(k)))
            push(prod)
    elif kind == "poseidon":
        rc = pymodel.round_constants()

        def sbox(x):
            x2 = B.mul(x, x)
            x4 = B.mul(x2, x2)
            x3 = B.mul(x2, x)
            return B.mul(x3, x4)

        def mds_then_constants(st, next_round):
            o = B.mds(st)
            if next_round >= 0:
                o = [B.add(o[i], B.lift(rc[12 * next_round + i])) for i in range(12)]
            return o
        swap = wire(24)
        push(B.mul(swap, B.sub(swap, one)))
        st = [None] * 12
        for i in range(4):
            lhs, rhs, delta = wire(i), wire(i + 4), wire(25 + i)
            push(B.sub(B.mul(swap, B.sub(rhs, lhs)), delta))
            st[i] = B.add(lhs, delta)
            st[i + 4] = B.sub(rhs, delta)
        for i in range(8, 12):
            st[i] = wire(i)
        st = [B.add(st[i], B.lift(rc[i])) for i in range(12)]
        rnd = 0
        for r in range(4):
            if r != 0:
                for i in range(12):
                    w_in = wire(29 + 12 * (r - 1) + i)
                    push(B.sub(st[i], w_in))
                    st[i] = w_in
            st = mds_then_constants([sbox(x) for x in st], rnd + 1)
            rnd += 1
        for r in range(22):
            w_in = wire(65 + r)
            push(B.sub(st[0], w_in))
            st[0] = sbox(w_in)
            st = mds_then_constants(st, rnd + 1)
            rnd += 1
        for r in range(4):
            for i in range(12):
                w_in = wire(87 + 12 * r + i)
                push(B.sub(st[i], w_in))
                st[i] = w_in
            st = mds_then_constants([sbox(x) for x in st], rnd + 1 if rnd + 1 < 30 else -1)
            rnd += 1
        for i in range(12):
            push(B.sub(wire(12 + i), st[i]))
    elif kind == "poseidon_mds":
        ins = [walg(2 * i) for i in range(12)]
        ma, mb = B.mds([x[0] for x in ins]), B.mds([x[1] for x in ins])
        for r in range(12):
            push_alg(asub(walg(2 * (12 + r)), (ma[r], mb[r])))
    elif kind == "arithmetic_ext":
        c0, c1 = const(0), const(1)
        for i in range(g.p0):
            m0, m1, addend, o = walg(8 * i), walg(8 * i + 2), walg(8 * i + 4), walg(8 * i + 6)
            push_alg(asub(o, aadd(ascale(amul(m0, m1), c0), ascale(addend, c1))))
    elif kind == "mul_ext":
        c0 = const(0)
        for i in range(g.p0):
            m0, m1, o = walg(6 * i), walg(6 * i + 2), walg(6 * i + 4)
            push_alg(asub(o, ascale(amul(m0, m1), c0)))
    elif kind in ("reducing", "reducing_ext"):
        n = g.p0
        ext = kind == "reducing_ext"
        alpha, acc = walg(2), walg(4)
        start_accs = 6 + 2 * n if ext else 6 + n
        for i in range(n):
            nxt = walg(0 if i == n - 1 else start_accs + 2 * i)
            c = amul(acc, alpha)
            if ext:
                c = aadd(c, walg(6 + 2 * i))
            else:
                c = (B.add(c[0], wire(6 + i)), c[1])
            push_alg(asub(c, nxt))
            acc = nxt
    elif kind == "random_access":
        bits, copies, extra = g.p0, g.p1, g.p2
        vec = 1 << bits
        routed = (2 + vec) * copies + extra
        for c in range(copies):
            base, bit0 = (2 + vec) * c, routed + c * bits
            bit = [wire(bit0 + b) for b in range(bits)]
            for b in range(bits):
                push(B.mul(bit[b], B.sub(bit[b], one)))
            idx = B.lift(0)
            for b in reversed(range(bits)):
                idx = B.add(B.add(idx, idx), bit[b])
            push(B.sub(idx, wire(base)))
            items = [wire(base + 2 + i) for i in range(vec)]
            for b in range(bits):
                items = [B.add(items[2 * i], B.mul(bit[b], B.sub(items[2 * i + 1], items[2 * i]))) for i in range(len(items) // 2)]
            push(B.sub(items[0], wire(base + 1)))
        for i in range(extra):
            push(B.sub(const(i), wire((2 + vec) * copies + i)))
    elif kind == "exponentiation":
        n = g.p0
        base = wire(0)
        prev = one
        for i in range(n):
            sq = one if i == 0 else B.mul(prev, prev)
            bit = wire(1 + (n - 1 - i))
            computed = B.mul(sq, B.add(B.mul(bit, base), B.sub(one, bit)))
            cur = wire(2 + n + i)
            push(B.sub(computed, cur))
            prev = cur
        push(B.sub(wire(1 + n), prev))
    elif kind == "coset_interpolation":
        bits, degree = g.p0, g.p1
        points = 1 << bits
        n_inter = (points - 2) // (degree - 1)
        start_point = 1 + 2 * points
        start_value, start_inter = start_point + 2, start_point + 4
        start_shifted = start_inter + 4 * n_inter
        dom, wts = barycentric(bits)
        shift = wire(0)
        point, shifted = walg(start_point), walg(start_shifted)
        push_alg(asub(point, ascale(shifted, shift)))
        ev, prod = (B.lift(0), B.lift(0)), (B.lift(1), B.lift(0))

        def partial(frm, to, ev, prod):
            for i in range(frm, to):
                val = ascalec(walg(1 + 2 * i), wts[i])
                term = (B.sub(shifted[0], B.lift(dom[i])), shifted[1])
                ev = aadd(amul(ev, term), amul(val, prod))
                prod = amul(prod, term)
            return ev, prod
        ev, prod = partial(0, min(degree, points), ev, prod)
        for i in range(n_inter):
            ie, ip = walg(start_inter + 2 * i), walg(start_inter + 2 * (n_inter + i))
            push_alg(asub(ie, ev))
            push_alg(asub(ip, prod))
            frm = 1 + (degree - 1) * (i + 1)
            ev, prod = partial(frm, min(frm + degree - 1, points), ie, ip)
        push_alg(asub(walg(start_value), ev))
    else:
        assert kind == "noop", kind
    assert len(out) == g.num_constraints, (kind, len(out), g.num_constraints)
    return out


def gate_terms(B, gates, api, local_constants, local_wires, pi_hash):
    """evaluate_gate_constraints: sum over the gates of filter * constraint, index-wise -> list of num_gate_constraints elements.
    local_constants: the constants openings (selectors first); filter = prod_{i in group, i != index} (i - s) * (UNUSED - s)"""
    total = [None] * gates.num_gate_constraints
    ns = gates.num_selectors
    for g in gates:
        kind = api.GATE_KINDS[g.kind]
        if g.num_constraints == 0:
            continue
        cons = gate_constraints(B, kind, g, lambda i: local_wires[i], lambda i: local_constants[ns + i], lambda i: pi_hash[i])
        sel = local_constants[g.selector_index]
        f = None
        for i in list(range(g.group_start, g.group_end)) + ([api.UNUSED_SELECTOR] if ns > 1 else []):
            if i == g.index:
                continue
            term = B.sub(B.lift(i), sel)
            f = term if f is None else B.mul(f, term)
        for k, c in enumerate(cons):
            t = c if f is None else B.mul(f, c)
            total[k] = t if total[k] is None else B.add(total[k], t)
    return [B.lift(0) if t is None else t for t in total]


# ================= the shape of the proofs a circuit's verifier accepts (CommonCircuitData) =================
class Shape:
    """What plonky2 keeps in CommonCircuitData for the inner circuit: degree, column counts, gate set / selectors, FRI parameters,
    number of public inputs -- and the flat layout of a step proof as vpbs_prove_step emits it (caps, openings, FriProof words)."""

    def __init__(self, api, log_n, n_pi, n_routed=N_ROUTED, n_wires=135, nc=2, deg=8, rate_bits=3, cap_height=4, pow_bits=16, n_queries=28):
        self.api, self.log_n, self.n_pi = api, log_n, n_pi
        self.gates = api.GateSet(GATE_SPEC)
        self.n_constants = self.gates.num_selectors + NUM_CONSTS
        self.n_routed, self.n_wires, self.nc, self.deg = n_routed, n_wires, nc, deg
        self.rate_bits, self.cap_height, self.pow_bits, self.n_queries = rate_bits, cap_height, pow_bits, n_queries
        self.n_chunks = -(-n_routed // deg)
        self.ncols = [self.n_constants + n_routed, n_wires, nc * self.n_chunks, nc << rate_bits]
        self.total_cols = sum(self.ncols)
        self.log_lde = log_n + rate_bits
        self.arity_bits, d = [], log_n                       # FriReductionStrategy::ConstantArityBits(4, 5)
        while d > 5 and d + rate_bits - 4 >= cap_height:
            self.arity_bits.append(4)
            d -= 4
        self.final_len = 1 << d
        self.cap_len = 1 << cap_height
        cap_words = 4 * self.cap_len
        self.caps_words = 3 * cap_words
        self.openings_words = 2 * (self.total_cols + nc)
        # FriProof: commit-phase caps, query rounds, final polynomial, pow witness
        self.nsib0 = self.log_lde - cap_height
        q = sum(c + 4 * self.nsib0 for c in self.ncols)
        lg = self.log_lde
        self.nsib = []
        for ab in self.arity_bits:
            lg -= ab
            self.nsib.append(lg - cap_height)
            q += (2 << ab) + 4 * (lg - cap_height)
        self.query_words = q
        self.fri_words = len(self.arity_bits) * cap_words + n_queries * q + 2 * self.final_len + 1
        self.proof_words = self.caps_words + self.openings_words + self.fri_words
        self.vk_words = 4 + cap_words

    def ext_pairs(self):
        """offsets (of the first word) of the pairs of the flat proof that are ExtensionTargets in plonky2's ProofTarget -- the openings, the
        16 evaluations of every FriQueryStep, the final polynomial's coefficients; every other word is a base Target (cap and sibling
        hashes, the initial trees' leaf values, the proof-of-work witness)"""
        out = [self.caps_words + 2 * j for j in range(self.total_cols + self.nc)]
        fri0 = self.caps_words + self.openings_words
        q_at = len(self.arity_bits) * 4 * self.cap_len
        for qi in range(self.n_queries):
            w = q_at + qi * self.query_words + sum(c + 4 * self.nsib0 for c in self.ncols)
            for r, ab in enumerate(self.arity_bits):
                out += [fri0 + w + 2 * i for i in range(1 << ab)]
                w += (2 << ab) + 4 * self.nsib[r]
            assert w == q_at + (qi + 1) * self.query_words
        final_at = self.fri_words - 1 - 2 * self.final_len
        out += [fri0 + final_at + 2 * i for i in range(self.final_len)]
        return out

    def flat_proof(self, proof):
        """the product's proof dict -> the flat word list the proof targets take"""
        out = np.concatenate([np.asarray(proof[k], dtype=np.uint64).reshape(-1) for k in ("caps", "openings", "fri")])
        assert out.size == self.proof_words
        return out


def hash_pad(words=()):
    """Hasher::hash_pad (plonk/config.rs): pad10*1 to a multiple of the rate 8, then hash_no_pad"""
    padded = [int(x) for x in words] + [1]
    while (len(padded) + 1) % 8:
        padded.append(0)
    return pymodel.hash_no_pad(padded + [1])


def circuit_digest(cs_cap, log_n, domain_separator=True):
    """verifier_only.circuit_digest as CircuitBuilder::build derives it (plonk/circuit_builder.rs): hash_no_pad(constants_sigmas_cap ||
    hash_pad(domain_separator = []) || degree_bits); domain_separator=False: the formula without the separator (the other position of
    vpbs_compat.digest_domain_separator)"""
    sep = [int(x) for x in hash_pad()] if domain_separator else []
    return np.array(pymodel.hash_no_pad([int(x) for x in np.asarray(cs_cap).reshape(-1)] + sep + [log_n]), np.uint64)


class CircuitChallenger:
    """iop/challenger.rs RecursiveChallenger: the duplex sponge of the transcript over PoseidonGate rows (overwrite mode, outputs popped
    from the end of the rate part) -- the schedule is static, so this is a straight-line circuit."""

    def __init__(self, cb):
        self.cb = cb
        self.state = [cb.zero()] * 12
        self.inp, self.out = [], []

    def _duplex(self):
        self.state = self.inp + self.state[len(self.inp):]
        self.inp = []
        self.state = self.cb.permute(self.state)
        self.out = self.state[:8]

    def observe(self, targets):
        for t in targets:
            self.out = []
            self.inp.append(t)
            if len(self.inp) == 8:
                self._duplex()

    def get(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def get_ext(self):
        a = self.get()
        return (a, self.get())


def verify_proof(cb, shape, proof, pis, vk_digest, vk_cap):
    """recursion/recursive_verifier.rs verify_proof for a step proof (flat targets `proof`, public-input targets `pis`) against the verifier
    data (vk_digest [4], vk_cap [16][4]): every check of csrc/verifier.hip vpbs_verify_step, in circuit.  Unsatisfiable iff the proof is bad."""
    S, B, api = shape, CircuitBackend(cb), shape.api
    assert len(proof) == S.proof_words and len(pis) == S.n_pi
    nc, cap_words = S.nc, 4 * S.cap_len
    caps = [[proof[o * cap_words + 4 * i:o * cap_words + 4 * i + 4] for i in range(S.cap_len)] for o in range(3)]
    op_flat = proof[S.caps_words:S.caps_words + S.openings_words]
    op = [(op_flat[2 * j], op_flat[2 * j + 1]) for j in range(S.total_cols + nc)]
    fri = proof[S.caps_words + S.openings_words:]
    # ---- transcript (plonk/get_challenges.rs) ----
    pi_hash = cb.hash_no_pad(pis)
    ch = CircuitChallenger(cb)
    ch.observe(vk_digest)
    ch.observe(pi_hash)
    ch.observe([t for h in caps[0] for t in h])
    betas = [ch.get() for _ in range(nc)]
    gammas = [ch.get() for _ in range(nc)]
    ch.observe([t for h in caps[1] for t in h])
    alphas = [ch.get() for _ in range(nc)]
    ch.observe([t for h in caps[2] for t in h])
    zeta = ch.get_ext()
    ch.observe(op_flat)
    fri_alpha = ch.get_ext()
    n_rounds = len(S.arity_bits)
    fri_caps, fri_betas = [], []
    for r in range(n_rounds):
        words = fri[r * cap_words:(r + 1) * cap_words]
        fri_caps.append([words[4 * i:4 * i + 4] for i in range(S.cap_len)])
        ch.observe(words)
        fri_betas.append(ch.get_ext())
    final_at = S.fri_words - 1 - 2 * S.final_len
    final_words = fri[final_at:final_at + 2 * S.final_len]
    final = [(final_words[2 * i], final_words[2 * i + 1]) for i in range(S.final_len)]
    pow_witness = fri[S.fri_words - 1]
    ch.observe(final_words)
    ch.observe([pow_witness])
    pow_response = ch.get()
    x_challenges = [ch.get() for _ in range(S.n_queries)]

    # ---- vanishing identity at zeta (plonk/vanishing_poly.rs eval_vanishing_poly_circuit, verifier.rs) ----
    n_cs, n_w = S.ncols[0], S.ncols[1]
    cs_z, wires_z = op[:n_cs], op[n_cs:n_cs + n_w]
    zs_pp_z = op[n_cs + n_w:n_cs + n_w + S.ncols[2]]
    quot_z = op[n_cs + n_w + S.ncols[2]:S.total_cols]
    zs_next_z = op[S.total_cols:]
    sig_z, pps_z = cs_z[S.n_constants:], zs_pp_z[nc:]
    one = cb.one_ext()
    zeta_n = cb.exp_pow2_ext(zeta, S.log_n)
    z_h = cb.sub_ext(zeta_n, one)
    l0 = cb.div_ext(z_h, cb.mulc_ext(cb.sub_ext(zeta, one), 1 << S.log_n))
    terms = [cb.mul_ext(l0, cb.sub_ext(zs_pp_z[c], one)) for c in range(nc)]
    num_prods = S.n_chunks - 1
    k_is = [pow(7, j, P) for j in range(S.n_routed)]
    for c in range(nc):
        beta_e, gamma_e = cb.to_ext(betas[c]), cb.to_ext(gammas[c])
        bz = cb.mul_ext(beta_e, zeta)
        for kk in range(S.n_chunks):
            num = den = None
            for j in range(kk * S.deg, min((kk + 1) * S.deg, S.n_routed)):
                fn = cb.add_ext(cb.mul_add_ext(cb.cext(k_is[j]), bz, wires_z[j]), gamma_e)
                fd = cb.add_ext(cb.mul_add_ext(beta_e, sig_z[j], wires_z[j]), gamma_e)
                num = fn if num is None else cb.mul_ext(num, fn)
                den = fd if den is None else cb.mul_ext(den, fd)
            prev = zs_pp_z[c] if kk == 0 else pps_z[c * num_prods + kk - 1]
            nxt = zs_next_z[c] if kk == num_prods else pps_z[c * num_prods + kk]
            terms.append(cb.mul_sub_ext(prev, num, cb.mul_ext(nxt, den)))
    terms += gate_terms(B, S.gates, api, cs_z, wires_z, [cb.to_ext(h) for h in pi_hash])
    chunks = 1 << S.rate_bits
    for a in range(nc):
        alpha_e = cb.to_ext(alphas[a])
        acc = cb.zero_ext()
        for t in reversed(terms):
            acc = cb.mul_add_ext(acc, alpha_e, t)                       # reduce_with_powers
        q = cb.zero_ext()
        for m in reversed(range(chunks)):
            q = cb.mul_add_ext(q, zeta_n, quot_z[a * chunks + m])
        cb.connect_ext(acc, cb.mul_ext(z_h, q))

    # ---- FRI (fri/recursive_verifier.rs) ----
    cb.range_check(pow_response, 64 - S.pow_bits)                        # fri_verify_proof_of_work: leading zeros
    reduced0 = cb.reduce_ext(op[:S.total_cols], fri_alpha)               # PrecomputedReducedOpenings
    reduced1 = cb.reduce_ext(op[S.total_cols:], fri_alpha)
    zeta_next = cb.mulc_ext(zeta, pymodel.root_of_unity(S.log_n))
    alpha_nc = fri_alpha
    for _ in range(nc - 1):
        alpha_nc = cb.mul_ext(alpha_nc, fri_alpha)
    alpha_nc = cb.mul_ext(alpha_nc, fri_alpha) if nc == 0 else alpha_nc
    oracle_caps = [vk_cap] + caps
    w_lde = pymodel.root_of_unity(S.log_lde)
    g_arity_inv = {ab: inv(pymodel.root_of_unity(ab)) for ab in set(S.arity_bits)}

    def merkle_verify(leaf, index_bits, cap, siblings):
        """hash/merkle_proofs.rs verify_merkle_proof_to_cap_with_cap_index"""
        cur = cb.hash_or_noop(leaf)
        zero = cb.zero()
        for k, sib in enumerate(siblings):
            cur = cb.permute(cur + sib + [zero] * 4, index_bits[k])[:4]
        cap_bits = index_bits[len(siblings):len(siblings) + S.cap_height]
        cap_index = cb.le_sum(cap_bits)
        entry = cb.random_access_many(cap_index, [[cap[i][e] for i in range(S.cap_len)] for e in range(4)])
        for x, y in zip(cur, entry):
            cb.connect(x, y)

    q_at = n_rounds * cap_words
    for qi in range(S.n_queries):
        w = q_at + qi * S.query_words
        bits = cb.split_le(x_challenges[qi], 64)[:S.log_lde]             # low bits of the challenge = x_index
        leaves = []
        for o in range(4):                                               # fri_verify_initial_proof
            leaf = fri[w:w + S.ncols[o]]
            w += S.ncols[o]
            sibs = [fri[w + 4 * k:w + 4 * k + 4] for k in range(S.nsib0)]
            w += 4 * S.nsib0
            merkle_verify(leaf, bits, oracle_caps[o], sibs)
            leaves.append(leaf)
        # subgroup_x = g * w^bitrev(x_index)
        x = cb.mul(cb.constant(7), cb.exp_from_bits_const_base(w_lde, list(reversed(bits))))
        xe = cb.to_ext(x)
        # fri_combine_initial
        acc0 = cb.reduce_base([t for leaf in leaves for t in leaf], fri_alpha)
        s = cb.div_ext(cb.sub_ext(acc0, reduced0), cb.sub_ext(xe, zeta))
        acc1 = cb.reduce_base(leaves[2][:nc], fri_alpha)
        s1 = cb.div_ext(cb.sub_ext(acc1, reduced1), cb.sub_ext(xe, zeta_next))
        old_eval = cb.mul_add_ext(s, alpha_nc, s1)
        for r, ab in enumerate(S.arity_bits):
            arity = 1 << ab
            ev_words = fri[w:w + 2 * arity]
            w += 2 * arity
            evals = [(ev_words[2 * i], ev_words[2 * i + 1]) for i in range(arity)]
            sibs = [fri[w + 4 * k:w + 4 * k + 4] for k in range(S.nsib[r])]
            w += 4 * S.nsib[r]
            within_bits, coset_bits = bits[:ab], bits[ab:]
            within = cb.le_sum(within_bits)
            picked = cb.random_access_many(within, [[e[0] for e in evals], [e[1] for e in evals]])
            cb.connect_ext((picked[0], picked[1]), old_eval)
            # compute_evaluation: interpolate on the coset and evaluate at beta_r
            start = cb.mul(x, cb.exp_from_bits_const_base(g_arity_inv[ab], list(reversed(within_bits))))
            ys = [evals[int(format(i, "0%db" % ab)[::-1], 2)] for i in range(arity)]
            old_eval = cb.interpolate_coset(start, ys, fri_betas[r])
            merkle_verify(ev_words, coset_bits, fri_caps[r], sibs)
            x = cb.exp_pow2(x, ab)
            bits = coset_bits
        cb.connect_ext(cb.reduce_ext(final, cb.to_ext(x)), old_eval)     # final_poly.eval(subgroup_x)
        assert w == q_at + (qi + 1) * S.query_words
    return {"pi_hash": pi_hash, "betas": betas, "gammas": gammas, "alphas": alphas, "zeta": zeta}


class VerifierOnlyCircuit:
    """A circuit whose only content is `verify_proof` of one inner proof (plonky2's recursion test shape): the unit test of the gadget."""

    def __init__(self, api, inner_shape, min_log_n=5):
        cb = ExtBuilder()
        self.shape = inner_shape
        self.proof = cb.virtuals(inner_shape.proof_words)
        self.pis = cb.virtuals(inner_shape.n_pi)
        self.vk_digest = cb.virtuals(4)
        self.vk_cap = [cb.virtuals(4) for _ in range(inner_shape.cap_len)]
        self.challenges = verify_proof(cb, inner_shape, self.proof, self.pis, self.vk_digest, self.vk_cap)
        cb.register_public_inputs(self.pis[:4])
        self.built = cb.build(api, min_log_n)

    def presets(self, flat_proof, pis, digest, cap):
        a = dict(zip(self.proof, (int(v) for v in flat_proof)))
        a.update(zip(self.pis, (int(v) for v in pis)))
        a.update(zip(self.vk_digest, (int(v) for v in digest)))
        for tgt, h in zip(self.vk_cap, np.asarray(cap).reshape(-1, 4)):
            a.update(zip(tgt, (int(v) for v in h)))
        return self.built.presets(a)


# ================= the cyclic step circuit (ivc_based_vpbs.rs:159-275) =================
def cyclic_n_pi(N, K):
    """public inputs of the cyclic step circuit: acc_init [K N], counter, current accumulator [K N], two chain hashes, then the verifier
    data (circuit digest [4], constants/sigmas cap [16][4]) -- add_verifier_data_public_inputs (ivc_based_vpbs.rs:196-214)"""
    return 2 * K * N + 1 + 8 + 4 + 64


class DummyCircuit:
    """recursion/dummy_circuit.rs: a circuit with the cyclic circuit's shape (degree, gate set, number of public inputs) and nothing else; its
    proof fills the inner-proof slot of the base step (cyclic_base_proof, ivc_based_vpbs.rs:292-299)"""

    def __init__(self, api, log_n, n_pi):
        cb = ExtBuilder()
        self.pis = cb.virtuals(n_pi)
        cb.register_public_inputs(self.pis)
        self.built = cb.build(api, log_n)
        assert self.built.log_n == log_n

    def witness(self, public_inputs):
        return self.built.circuit.generate_witness(self.built.presets(dict(zip(self.pis, (int(v) for v in public_inputs)))))


class CyclicStepCircuit:
    """verified_pbs's circuit: build_step_circuit + the verifier of the previous step's proof of THIS circuit.  log_n is the degree the
    circuit is built for (its own proofs, hence the proofs it verifies, have that degree): the constructor fails if the rows do not fit."""

    def __init__(self, api, N, K, ELL, LOGB, n_lwe, ring, log_n):
        cb = ExtBuilder()
        self.step = step = sc.StepCircuit.__new__(sc.StepCircuit)
        step.logic(cb, N, K, ELL, LOGB, n_lwe, ring)
        self.params = (N, K, ELL, LOGB, n_lwe)
        # verifier data as public inputs (add_verifier_data_public_inputs, :214)
        self.vk_digest = cb.virtuals(4)
        self.vk_cap = [cb.virtuals(4) for _ in range(16)]
        cb.register_public_inputs(self.vk_digest)
        for h in self.vk_cap:
            cb.register_public_inputs(h)
        n_pi = len(cb.public_inputs)
        assert n_pi == cyclic_n_pi(N, K)
        self.shape = shape = Shape(api, log_n, n_pi)
        # the inner proof and its public inputs (:217-229)
        self.proof = cb.virtuals(shape.proof_words)
        self.inner_pis = ip = cb.virtuals(n_pi)
        kn = K * N
        inner_acc_init, inner_counter = ip[:kn], ip[kn]
        inner_acc = [ip[kn + 1 + p * N:kn + 1 + (p + 1) * N] for p in range(K)]
        inner_bsk_hash, inner_lwe_hash = ip[2 * kn + 1:2 * kn + 5], ip[2 * kn + 5:2 * kn + 9]
        for a, b in zip([t for p in step.acc_init for t in p], inner_acc_init):     # :231-237
            cb.connect(a, b)
        self.condition = cond = cb.virtual()                                         # add_virtual_bool_target_safe (:240)
        cb.connect(cb.mul_sub(cond, cond, cond), cb.zero())
        actual_acc_in = sc.glwe_select(cb, cond, inner_acc, step.acc_init)           # :241-248
        for a, b in zip([t for p in step.acc_in for t in p], [t for p in actual_acc_in for t in p]):
            cb.connect(a, b)
        zero = cb.zero()
        for a, b in zip(step.bsk_hash_in, inner_bsk_hash):                           # :250-266
            cb.connect(a, cb.select(cond, b, zero))
        for a, b in zip(step.lwe_hash_in, inner_lwe_hash):
            cb.connect(a, cb.select(cond, b, zero))
        cb.connect(step.counter, cb.mul_add(cond, inner_counter, cb.one()))          # :268-269
        # conditionally_verify_cyclic_proof_or_dummy (:271-277; plonky2 recursion/cyclic_recursion.rs + conditional_recursive_verifier.rs):
        #  * the verifier data carried by the cyclic proof's public inputs are this circuit's own (connect_hashes / connect_merkle_caps,
        #    unconditionally: the base proof carries them too);
        #  * a SECOND proof slot -- the dummy circuit's proof with its public inputs and verifier data, free targets that plonky2's
        #    DummyProofGenerator fills (dummy_proof_and_vk) and the driver presets here;
        #  * select_proof_with_pis: EVERY word of the proof and of its public inputs is selected between the two slots by `condition`, the
        #    verifier data between own and dummy (select_hash / select_cap), and ONE verifier checks the selection (verify_proof).
        own_vk = self.vk_digest + [t for h in self.vk_cap for t in h]
        for a, b in zip(own_vk, ip[n_pi - 68:]):
            cb.connect(a, b)
        self.dummy_proof = cb.virtuals(shape.proof_words)
        self.dummy_pis = cb.virtuals(n_pi)
        self.dummy_vk = cb.virtuals(68)
        sel_proof = [None] * shape.proof_words
        for at in shape.ext_pairs():      # ExtensionTargets of the ProofTarget go through select_ext, the rest through select
            sel_proof[at], sel_proof[at + 1] = cb.select_ext(cond, (self.proof[at], self.proof[at + 1]), (self.dummy_proof[at], self.dummy_proof[at + 1]))
        for at in range(shape.proof_words):
            if sel_proof[at] is None:
                sel_proof[at] = cb.select(cond, self.proof[at], self.dummy_proof[at])
        sel_pis = [cb.select(cond, a, b) for a, b in zip(ip, self.dummy_pis)]
        sel = [cb.select(cond, a, b) for a, b in zip(own_vk, self.dummy_vk)]
        verify_proof(cb, shape, sel_proof, sel_pis, sel[:4], [sel[4 + 4 * i:8 + 4 * i] for i in range(16)])
        self.built = cb.build(api, log_n)
        if self.built.log_n != log_n:
            raise ValueError("the cyclic circuit needs %d rows: it does not fit degree 2^%d" % (self.built.used_rows, log_n))
        self.targets = (self.proof + ip + [cond] + step.ggsw_flat + [step.mask] + own_vk + self.dummy_vk + self.dummy_proof + self.dummy_pis)
        self.positions = [self.built.pos(t) for t in self.targets]

    def values(self, flat_proof, inner_pis, condition, ggsw_flat, mask, own_vk, dummy_vk, dummy_flat_proof, dummy_pis=None):
        """the PartialWitness of one step in the order of self.targets / self.positions (ivc_based_vpbs.rs:283-299, 314-330, 345-361); the
        last two entries are what plonky2's DummyProofGenerator sets: the dummy circuit's proof (of all-zero public inputs) and those inputs"""
        dummy_pis = np.zeros(len(self.dummy_pis), np.uint64) if dummy_pis is None else np.asarray(dummy_pis, np.uint64)
        v = np.concatenate([np.asarray(flat_proof, np.uint64), np.asarray(inner_pis, np.uint64), np.array([condition], np.uint64),
                            np.asarray(ggsw_flat, np.uint64).reshape(-1), np.array([int(mask) % P], np.uint64), np.asarray(own_vk, np.uint64),
                            np.asarray(dummy_vk, np.uint64), np.asarray(dummy_flat_proof, np.uint64), dummy_pis])
        assert v.size == len(self.targets)
        return v

    def public_inputs(self, wires):
        return self.built.values(wires, self.built.public_inputs)


def vk_words(cs_cap, log_n, domain_separator=True):
    """verifier data as the 68 words the public inputs carry: circuit digest, then the constants/sigmas cap"""
    return np.concatenate([circuit_digest(cs_cap, log_n, domain_separator), np.asarray(cs_cap, np.uint64).reshape(-1)])
