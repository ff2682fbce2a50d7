"""TEST INFRASTRUCTURE: a minimal circuit builder and the reference's step circuit without the recursive verifier.

The product takes a finished circuit description (vpbs_circuit: gate per row, constants, copy constraints, gadget generators) and a
PartialWitness; building that description is the caller's job (the reference does it with plonky2's CircuitBuilder).  To exercise the
product's witness generation, prover and verifier on the computation the reference actually proves, this file restates

  * the parts of plonky2 0.2.0's CircuitBuilder the step circuit needs (plonk/circuit_builder.rs: constants, arithmetic-gate slot
    packing, public inputs and their hash row; gadgets/arithmetic.rs: arithmetic + arithmetic_special_cases, add/sub/mul/neg/mul_add/
    mul_sub/mul_const_add, is_equal, not; gadgets/select.rs: select; gadgets/split_join.rs: split_le; gadgets/hash.rs + hashing.rs:
    hash_n_to_hash_no_pad over PoseidonGate rows, overwrite-mode sponge), and
  * the reference's gadgets and `build_step_circuit` (/root/reference/src/vtfhe/ivc_based_vpbs.rs:80-155; mod.rs:52-117 poly_select /
    glwe_select / rotate_poly / rotate_glwe; glwe_poly.rs:17-50 plus_or_minus / decompose, :132-166 rotate / decompose;
    glev_ct.rs:92-110 mul; ggsw_ct.rs:11-28,98-112 glwe_add_many / external_product; ntt/mod.rs:12-66; vec_arithmetic.rs:6-63).

Row placement is this builder's own (it is not claimed to equal plonky2's: parity with the crate is unpinned); the
gate semantics, the gadget arithmetic and the public-input order follow the sources above.  Nothing here is product code.
"""
import numpy as np

P = 0xFFFFFFFF00000001
N_WIRES, N_ROUTED = 135, 80
ARITH_OPS, BASE_SUM_LIMBS, NUM_CONSTS = 20, 63, 2
GATE_SPEC = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon"]


class Builder:
    gate_spec = GATE_SPEC          # the circuit's gate set (circuitgen/cyclic_circuit.py widens it to the 14 standard gates)

    def __init__(self):
        self.parent, self.t_row, self.t_col = [], [], []      # targets: ids; wire targets carry (row, column), virtual ones (-1, -1)
        self.rows, self.row_consts = [], []                    # gate kind and gate constants per row
        self.arith_free, self.const_free = {}, None            # (c0, c1) -> (row, next op) ; (row, next slot)
        self.const_target, self.const_of = {}, {}              # value -> target ; target -> value
        self.generators, self.public_inputs = [], []

    # ---- targets and copy constraints ----
    def _new(self, row=-1, col=-1):
        self.parent.append(len(self.parent))
        self.t_row.append(row)
        self.t_col.append(col)
        return len(self.parent) - 1

    def virtual(self):
        return self._new()

    def virtuals(self, count):
        return [self._new() for _ in range(count)]

    def find(self, t):
        p = self.parent
        while p[t] != t:
            p[t] = p[p[t]]
            t = p[t]
        return t

    def connect(self, a, b):
        a, b = self.find(a), self.find(b)
        if a != b:
            self.parent[max(a, b)] = min(a, b)

    def _wire(self, row, col, source=None):
        """a fresh wire target, optionally copy-constrained to `source`"""
        t = self._new(row, col)
        if source is not None:
            self.parent[t] = source
        return t

    def add_row(self, kind, consts=()):
        self.rows.append(kind)
        self.row_consts.append(list(consts))
        return len(self.rows) - 1

    # ---- constants (ConstantGate rows, num_consts = 2) ----
    def constant(self, value):
        value %= P
        t = self.const_target.get(value)
        if t is None:
            if self.const_free is None:
                self.const_free = (self.add_row("constant", [0] * NUM_CONSTS), 0)
            row, slot = self.const_free
            self.row_consts[row][slot] = value
            t = self._wire(row, slot)
            self.const_free = (row, slot + 1) if slot + 1 < NUM_CONSTS else None
            self.const_target[value] = t
            self.const_of[t] = value
        return t

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    # ---- gadgets/arithmetic.rs ----
    def arithmetic(self, c0, x, y, c1, z):
        """c0 * x * y + c1 * z on an ArithmeticGate slot (after arithmetic_special_cases)"""
        c0 %= P
        c1 %= P
        zero, k = self.zero(), self.const_of
        xc, yc, zc = k.get(x), k.get(y), k.get(z)
        first_zero = c0 == 0 or x == zero or y == zero
        second_zero = c1 == 0 or z == zero
        first_const = 0 if first_zero else (xc * yc * c0 % P if xc is not None and yc is not None else None)
        second_const = 0 if second_zero else (zc * c1 % P if zc is not None else None)
        if first_const is not None and second_const is not None:
            return self.constant(first_const + second_const)
        if first_zero and c1 == 1:
            return z
        if second_zero:
            if xc is not None and xc * c0 % P == 1:
                return y
            if yc is not None and yc * c0 % P == 1:
                return x
        key = (c0, c1)
        slot = self.arith_free.get(key)
        if slot is None:
            slot = (self.add_row("arithmetic", [c0, c1]), 0)
        row, i = slot
        if i + 1 < ARITH_OPS:
            self.arith_free[key] = (row, i + 1)
        else:
            self.arith_free.pop(key, None)
        self._wire(row, 4 * i, x)
        self._wire(row, 4 * i + 1, y)
        self._wire(row, 4 * i + 2, z)
        return self._wire(row, 4 * i + 3)

    def mul(self, x, y):
        return self.arithmetic(1, x, y, 0, self.zero())

    def add(self, x, y):
        return self.arithmetic(1, x, self.one(), 1, y)

    def sub(self, x, y):
        return self.arithmetic(1, y, self.constant(P - 1), 1, x)

    def neg(self, x):
        return self.mul(x, self.constant(P - 1))

    def mul_add(self, x, y, z):
        return self.arithmetic(1, x, y, 1, z)

    def mul_sub(self, x, y, z):
        return self.arithmetic(1, x, y, P - 1, z)

    def mul_const_add(self, c, x, y):
        return self.arithmetic(c, x, self.one(), 1, y)

    def select(self, b, x, y):                                 # gadgets/select.rs: b x - (b y - y)
        return self.mul_sub(b, x, self.mul_sub(b, y, y))

    def is_equal(self, x, y):                                  # gadgets/arithmetic.rs is_equal + EqualityGenerator
        equal, inv = self.virtual(), self.virtual()
        not_equal = self.sub(self.one(), equal)
        self.generators.append(("equality", 0, [x, y], [equal, inv]))
        diff = self.sub(x, y)
        self.connect(not_equal, self.mul(diff, inv))
        self.connect(diff, self.mul(not_equal, diff))
        return equal

    # ---- gadgets/split_join.rs ----
    def split_le(self, integer, num_bits):
        k = -(-num_bits // BASE_SUM_LIMBS)
        sums, bits = [], []
        for _ in range(k):
            row = self.add_row("base_sum")
            sums.append(self._wire(row, 0))
            bits += [self._wire(row, 1 + i) for i in range(BASE_SUM_LIMBS)]
        zero = self.zero()
        for b in bits[num_bits:]:                               # gadgets/split_join.rs: `for b in bits.drain(num_bits..) { assert_zero(b) }`
            self.connect(b, zero)
        del bits[num_bits:]
        acc = self.zero()
        for s in reversed(sums):
            acc = self.mul_const_add(1 << BASE_SUM_LIMBS, acc, s)
        self.connect(acc, integer)
        self.generators.append(("wire_split", BASE_SUM_LIMBS, [integer], sums))
        return bits

    def le_sum(self, bits):                                    # the arithmetic branch (num_bits - 1 <= ops per gate)
        assert 0 < len(bits) <= ARITH_OPS + 1
        two, acc = self.constant(2), bits[-1]
        for b in reversed(bits[:-1]):
            acc = self.mul_add(two, acc, b)
        return acc

    # ---- gadgets/hash.rs, hash/hashing.rs: hash_n_to_hash_no_pad::<PoseidonHash> ----
    def hash_no_pad(self, inputs):
        zero = self.zero()
        state = [zero] * 12
        for at in range(0, len(inputs), 8):
            chunk = inputs[at:at + 8]
            row = self.add_row("poseidon")
            for i in range(12):
                self._wire(row, i, chunk[i] if i < len(chunk) else state[i])   # overwrite mode
            self._wire(row, 24, zero)                                           # swap = false
            state = [self._wire(row, 12 + i) for i in range(12)]
        return state[:4]

    def register_public_inputs(self, targets):
        self.public_inputs += list(targets)

    def fill_unused_slots(self):
        """hook: gates with several operations per row give their unused operations defined inputs before build()"""

    # ---- CircuitBuilder::build ----
    def build(self, api, min_log_n=5):
        """-> Built (circuit description for the product; target -> wire position map)"""
        pi_hash = self.hash_no_pad(self.public_inputs)
        row = self.add_row("public_input")
        for i in range(4):
            self._wire(row, i, pi_hash[i])
        zero = self.zero()
        for (c0, c1), (row, i) in list(self.arith_free.items()):   # unused operations compute 0 * 0 * c0 + c1 * 0
            for j in range(i, ARITH_OPS):
                for w in range(3):
                    self._wire(row, 4 * j + w, zero)
        self.arith_free.clear()
        self.fill_unused_slots()
        log_n = max(min_log_n, (len(self.rows) - 1).bit_length())
        n = 1 << log_n
        gates = api.GateSet(self.gate_spec)
        names = [k if isinstance(k, str) else k[0] for k in self.gate_spec]
        index = {kind: gates.by_kind(kind).index for kind in names}
        row_gate = np.full(n, index["noop"], np.uint32)
        row_gate[:len(self.rows)] = [index[k] for k in self.rows]
        constants = np.zeros((gates.num_selectors + NUM_CONSTS, n), np.uint64)
        for r, cs in enumerate(self.row_consts):
            for j, c in enumerate(cs):
                constants[gates.num_selectors + j, r] = c
        # copy-constraint classes: chain the routed wires of every class, remember one wire per class
        count = len(self.parent)
        root = [self.find(t) for t in range(count)]
        t_row, t_col = self.t_row, self.t_col
        last, copies = {}, []
        position = {}
        for t in range(count):
            if t_row[t] < 0:
                continue
            assert t_col[t] < N_ROUTED, "copy constraint on an advice wire"
            p = t_col[t] * n + t_row[t]
            r = root[t]
            q = last.get(r)
            if q is None:
                position[r] = p
            else:
                copies.append((q, p))
            last[r] = p
        generators = []
        for kind, p0, ins, outs in self.generators:
            generators.append((kind, p0, [divmod(position[root[t]], n) for t in ins], [divmod(position[root[t]], n) for t in outs]))
        copies = np.array(copies, dtype=np.uint32).reshape(-1, 2)
        circ = api.Circuit(gates, log_n, row_gate, constants, copies, generators=generators)
        constants[:gates.num_selectors] = circ.selector_columns()
        circ = api.Circuit(gates, log_n, row_gate, constants, copies, generators=generators)
        return Built(self, circ, gates, constants, root, position, len(self.rows))


class Built:
    def __init__(self, builder, circuit, gates, constants, root, position, used_rows):
        self.circuit, self.gates, self.constants, self.used_rows = circuit, gates, constants, used_rows
        self.log_n, self.n = circuit.log_n, circuit.n
        self._root, self._position = root, position
        self.public_inputs = list(builder.public_inputs)
        self.row_kinds = list(builder.rows)

    def pos(self, target):
        """(column, row) of a wire in the target's copy class"""
        return divmod(self._position[self._root[target]], self.n)

    def presets(self, assignment):
        """{target: value} -> PartialWitness for Circuit.generate_witness"""
        return {self.pos(t): int(v) % P for t, v in assignment.items()}

    def values(self, wires, targets):
        return [int(wires[self.pos(t)]) for t in targets]


# ---- the reference's gadgets ----
def poly_select(cb, control, left, right):                    # mod.rs:52-65
    return [cb.select(control, l, r) for l, r in zip(left, right)]


def glwe_select(cb, control, left, right):                    # mod.rs:67-78
    return [poly_select(cb, control, l, r) for l, r in zip(left, right)]


def poly_rotate(cb, poly, shift):                             # glwe_poly.rs:132-148
    n = len(poly)
    return [cb.neg(poly[n - shift + i]) if i < shift else poly[i - shift] for i in range(n)]


def rotate_poly(cb, poly, shift):                             # mod.rs:80-107
    log2n = (len(poly) - 1).bit_length() + 1
    bits = cb.split_le(shift, 64)
    polys = [poly_select(cb, bits[64 - log2n - 1], poly_rotate(cb, poly, 1), poly)]
    for log_shift, bit in enumerate(bits[64 - log2n:]):
        current = polys[-1]
        polys.append(poly_select(cb, bit, poly_rotate(cb, current, 1 << log_shift), current))
    return polys[-1]


def rotate_glwe(cb, glwe, shift):                             # mod.rs:109-117
    return [rotate_poly(cb, p, shift) for p in glwe]


def plus_or_minus(cb, b, x):                                  # glwe_poly.rs:17-24
    return cb.select(b, cb.neg(x), x)


def decompose(cb, x, num_limbs, logb):                        # glwe_poly.rs:28-50
    bits = cb.split_le(x, num_limbs * logb)
    sgn = bits[-1]
    bits_centered = cb.split_le(plus_or_minus(cb, sgn, x), num_limbs * logb)
    base = cb.constant(1 << logb)
    carry, out = cb.zero(), []
    for at in range(0, len(bits_centered), logb):
        limb = bits_centered[at:at + logb]
        k_w_carry = cb.add(cb.le_sum(limb), carry)
        carry = limb[-1]
        out.append(plus_or_minus(cb, sgn, cb.sub(k_w_carry, cb.mul(carry, base))))
    return out


def poly_decompose(cb, poly, num_limbs, logb):                # glwe_poly.rs:150-166
    per_coeff = [decompose(cb, x, num_limbs, logb) for x in poly]
    return [[d[i] for d in per_coeff] for i in range(num_limbs)]


def ntt_forward(cb, poly, roots):                             # ntt/mod.rs:12-40
    a, n = list(poly), len(poly)
    m = 1
    while m < n:
        t = n // (2 * m)
        for i in range(m):
            s = cb.constant(int(roots[m + i]))
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = a[j], cb.mul(a[j + t], s)
                a[j], a[j + t] = cb.add(u, v), cb.sub(u, v)
        m *= 2
    return a


def ntt_backward(cb, poly, invroots, ninv):                   # ntt/mod.rs:43-76
    a, n = list(poly), len(poly)
    m = n // 2
    while m >= 1:
        t = n // (2 * m)
        for i in range(m):
            s = cb.constant(int(invroots[m + i]))
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = a[j], a[j + t]
                a[j], a[j + t] = cb.add(u, v), cb.mul(cb.sub(u, v), s)
        m //= 2
    n_inv = cb.constant(int(ninv))
    return [cb.mul(g, n_inv) for g in a]


def vec_inner(cb, left, right):                               # vec_arithmetic.rs:6-63 (vec_mul, then vec_add_many from zero)
    acc = [cb.zero()] * len(left[0])
    for l, r in zip(left, right):
        prod = [cb.mul(x, y) for x, y in zip(l, r)]
        acc = [cb.add(a, b) for a, b in zip(acc, prod)]
    return acc


def glev_mul(cb, glev, poly, K, ELL, logb, ring):             # glev_ct.rs:92-110; glev[l][r] = polynomial r of GLWE l
    num_limbs = -(-64 // logb)
    limbs = poly_decompose(cb, poly, num_limbs, logb)
    limbs_hat = [ntt_forward(cb, limb, ring[0]) for limb in limbs[num_limbs - ELL:]]
    return [vec_inner(cb, limbs_hat, [glev[l][r] for l in range(ELL)]) for r in range(K)]


def glwe_add(cb, a, b):
    return [[cb.add(x, y) for x, y in zip(p, q)] for p, q in zip(a, b)]


def glwe_sub(cb, a, b):
    return [[cb.sub(x, y) for x, y in zip(p, q)] for p, q in zip(a, b)]


def external_product(cb, ggsw, glwe, K, ELL, logb, ring):     # ggsw_ct.rs:98-112
    muls = [glev_mul(cb, glev, poly, K, ELL, logb, ring) for poly, glev in zip(glwe, ggsw)]
    total = [[cb.zero()] * len(glwe[0]) for _ in range(K)]    # glwe_add_many (ggsw_ct.rs:11-28)
    for m in muls[:K - 1]:
        total = glwe_add(cb, total, m)
    return [ntt_backward(cb, p, ring[1], ring[2]) for p in glwe_sub(cb, muls[K - 1], total)]


class StepCircuit:
    """build_step_circuit (ivc_based_vpbs.rs:80-155) for ring dimension N, K polynomials per GLWE, ELL levels, base 2^LOGB and LWE
    dimension n_lwe.  Public inputs, in order: acc_init [K N], counter, current_acc_out [K N], current_bsk_hash_out [4],
    current_lwe_hash_out [4].  ring = (ROOTS, INVROOTS, NINV) of ntt/params_*.rs (the oracle's negacyclic_params)."""

    def __init__(self, api, N, K, ELL, LOGB, n_lwe, ring):
        cb = Builder()
        self.logic(cb, N, K, ELL, LOGB, n_lwe, ring)
        self.built = cb.build(api)

    def logic(self, cb, N, K, ELL, LOGB, n_lwe, ring):
        """build_step_circuit on the builder cb (ivc_based_vpbs.rs:80-155); registers the step's public inputs in the reference's order"""
        self.shape = (N, K, ELL, LOGB, n_lwe)
        self.acc_init = [cb.virtuals(N) for _ in range(K)]
        self.ggsw = [[[cb.virtuals(N) for _ in range(K)] for _ in range(ELL)] for _ in range(K)]   # [glev][glwe][poly][N]
        cb.register_public_inputs([t for p in self.acc_init for t in p])
        self.acc_in = [cb.virtuals(N) for _ in range(K)]
        self.counter = cb.virtual()
        cb.register_public_inputs([self.counter])
        first_step = cb.is_equal(self.counter, cb.one())
        last_step = cb.is_equal(self.counter, cb.constant(n_lwe + 2))
        self.mask = cb.virtual()
        first_negated_mask = cb.select(first_step, cb.neg(self.mask), self.mask)
        shifted = rotate_glwe(cb, self.acc_in, first_negated_mask)
        diff = glwe_sub(cb, shifted, self.acc_in)
        xprod_in = glwe_select(cb, last_step, self.acc_in, diff)
        xprod_out = external_product(cb, self.ggsw, xprod_in, K, ELL, LOGB, ring)
        cmux_out = glwe_add(cb, xprod_out, self.acc_in)
        cmux_or_exprod = glwe_select(cb, last_step, xprod_out, cmux_out)
        self.acc_out = glwe_select(cb, first_step, shifted, cmux_or_exprod)
        cb.register_public_inputs([t for p in self.acc_out for t in p])
        self.bsk_hash_in = cb.virtuals(4)
        self.ggsw_flat = [t for glev in self.ggsw for glwe in glev for poly in glwe for t in poly]
        self.bsk_hash_out = cb.hash_no_pad(self.bsk_hash_in + self.ggsw_flat)
        self.lwe_hash_in = cb.virtuals(4)
        self.lwe_hash_out = cb.hash_no_pad(self.lwe_hash_in + [self.mask])
        cb.register_public_inputs(self.bsk_hash_out)
        cb.register_public_inputs(self.lwe_hash_out)

    def witness(self, acc_init, ggsw_flat, acc_in, counter, mask, bsk_hash_in, lwe_hash_in):
        """PartialWitness of one step (ivc_based_vpbs.rs:283-299, 314-330, 345-361) -> wires via the PRODUCT's generators"""
        a = {}
        for targets, values in ((self.acc_init, acc_init), (self.acc_in, acc_in)):
            for tp, vp in zip(targets, values):
                a.update(zip(tp, vp))
        a.update(zip(self.ggsw_flat, ggsw_flat))
        a[self.counter], a[self.mask] = counter, mask
        a.update(zip(self.bsk_hash_in, bsk_hash_in))
        a.update(zip(self.lwe_hash_in, lwe_hash_in))
        return self.built.circuit.generate_witness(self.built.presets(a))

    def public_inputs(self, wires):
        return self.built.values(wires, self.built.public_inputs)
