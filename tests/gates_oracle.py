"""Gate-level test infrastructure (oracle side): an independent Python restatement of plonky2's gate bookkeeping
(derived parameters, Gate::id strings, CircuitBuilder's gate ordering, gates/selectors.rs selector_polynomials) and of
every gate's witness generator in pure-Python integers, plus the ctypes binding of oracle/gates.c.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.
"""
import ctypes as C

import numpy as np

import oracle as orc
import pymodel

P = orc.P
KINDS = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext",
         "reducing", "reducing_ext", "random_access", "exponentiation", "coset_interpolation"]
UNUSED_SELECTOR = 0xFFFFFFFF
FIELD = "plonky2_field::goldilocks_field::GoldilocksField"
# CircuitConfig::standard_recursion_config
NUM_WIRES, NUM_ROUTED, NUM_CONSTANTS, MAX_QUOTIENT_DEGREE_FACTOR = 135, 80, 2, 8
SELECTOR_MAX_DEGREE = MAX_QUOTIENT_DEGREE_FACTOR + 1  # CircuitBuilder::build passes quotient_degree_factor + 1 to selector_polynomials


class OrcGate(C.Structure):
    _fields_ = [("kind", C.c_uint), ("p0", C.c_uint), ("p1", C.c_uint), ("p2", C.c_uint),
                ("selector_index", C.c_uint), ("group_start", C.c_uint), ("group_end", C.c_uint), ("index", C.c_uint)]


def _bind():
    L = orc.lib()
    if getattr(L, "_gates_bound", False):
        return L
    sz, ui, U64P = C.c_size_t, C.c_uint, orc.U64P
    G = C.POINTER(OrcGate)
    L.orc_gate_terms_coset.restype, L.orc_gate_terms_coset.argtypes = C.c_int, [G, sz, sz, U64P, sz, U64P, sz, ui, U64P, U64P, sz, U64P]
    L.orc_gate_terms_zeta.restype, L.orc_gate_terms_zeta.argtypes = C.c_int, [G, sz, sz, U64P, sz, U64P, sz, U64P, U64P, sz, U64P]
    L.orc_gate_eval_row.restype, L.orc_gate_eval_row.argtypes = sz, [G, U64P, sz, U64P, sz, U64P, U64P]
    L._gates_bound = True
    return L


def barycentric_weights(bits):
    g = pymodel.root_of_unity(bits)
    dom = [pow(g, i, P) for i in range(1 << bits)]
    wts = []
    for i, xi in enumerate(dom):
        d = 1
        for j, xj in enumerate(dom):
            if j != i:
                d = d * (xi - xj) % P
        wts.append(pow(d, P - 2, P))
    return dom, wts


def default_params(kind, p0=0, p1=0, p2=0):
    """the *_from_config constructors under standard_recursion_config"""
    if kind == "constant":
        p0 = p0 or NUM_CONSTANTS
    elif kind == "arithmetic":
        p0 = p0 or NUM_ROUTED // 4
    elif kind == "base_sum":
        p1 = p1 or 2
        if not p0:
            limbs, v = 0, p1
            while v <= P:
                limbs, v = limbs + 1, v * p1
            p0 = min(limbs, NUM_ROUTED - 1)
    elif kind == "arithmetic_ext":
        p0 = p0 or NUM_ROUTED // 8
    elif kind == "mul_ext":
        p0 = p0 or NUM_ROUTED // 6
    elif kind == "reducing":
        p0 = p0 or min(NUM_ROUTED - 6, (NUM_WIRES - 4) // 3)
    elif kind == "reducing_ext":
        p0 = p0 or min((NUM_ROUTED - 6) // 2, (NUM_WIRES - 4) // 4)
    elif kind == "random_access":
        p0 = p0 or 4
        if not p1:
            vec = 1 << p0
            p1 = min(NUM_ROUTED // (2 + vec), NUM_WIRES // (2 + vec + p0))
            p2 = min(NUM_ROUTED - (2 + vec) * p1, NUM_CONSTANTS)
    elif kind == "exponentiation":
        p0 = p0 or min(NUM_ROUTED - 2, (NUM_WIRES - 2) // 2)
    elif kind == "coset_interpolation":
        p0 = p0 or 4
        if not p1:
            points = 1 << p0
            n_inter = (points - 2) // (MAX_QUOTIENT_DEGREE_FACTOR - 1)
            p1 = (points - 2) // (n_inter + 1) + 2
    return p0, p1, p2


def derived(kind, p0, p1, p2):
    """(degree, num_constraints, num_constants, num_wires)"""
    if kind == "noop":
        return 0, 0, 0, 0
    if kind == "constant":
        return 1, p0, p0, p0
    if kind == "public_input":
        return 1, 4, 0, 4
    if kind == "arithmetic":
        return 3, p0, 2, 4 * p0
    if kind == "base_sum":
        return p1, 1 + p0, 0, 1 + p0
    if kind == "poseidon":
        return 7, 123, 0, 135
    if kind == "poseidon_mds":
        return 1, 24, 0, 48
    if kind == "arithmetic_ext":
        return 3, 2 * p0, 2, 8 * p0
    if kind == "mul_ext":
        return 3, 2 * p0, 1, 6 * p0
    if kind == "reducing":
        return 2, 2 * p0, 0, 3 * p0 + 4
    if kind == "reducing_ext":
        return 2, 2 * p0, 0, 4 * p0 + 4
    if kind == "random_access":
        return p0 + 1, p1 * (p0 + 2) + p2, p2, (2 + (1 << p0)) * p1 + p2 + p1 * p0
    if kind == "exponentiation":
        return 4, p0 + 1, 0, 2 + 2 * p0
    if kind == "coset_interpolation":
        ni = ((1 << p0) - 2) // (p1 - 1)
        return p1, 2 * (2 + 2 * ni), 0, 1 + 2 * (1 << p0) + 4 + 4 * ni + 2
    raise ValueError(kind)


def gate_id(kind, p0, p1, p2):
    if kind == "noop":
        return "NoopGate"
    if kind == "constant":
        return "ConstantGate { num_consts: %d }" % p0
    if kind == "public_input":
        return "PublicInputGate"
    if kind == "arithmetic":
        return "ArithmeticGate { num_ops: %d }" % p0
    if kind == "base_sum":
        return "BaseSumGate { num_limbs: %d } + Base: %d" % (p0, p1)
    if kind == "poseidon":
        return "PoseidonGate(PhantomData<%s>)<WIDTH=12>" % FIELD
    if kind == "poseidon_mds":
        return "PoseidonMdsGate(PhantomData<%s>)<WIDTH=12>" % FIELD
    if kind == "arithmetic_ext":
        return "ArithmeticExtensionGate { num_ops: %d }" % p0
    if kind == "mul_ext":
        return "MulExtensionGate { num_ops: %d }" % p0
    if kind == "reducing":
        return "ReducingGate { num_coeffs: %d }" % p0
    if kind == "reducing_ext":
        return "ReducingExtensionGate { num_coeffs: %d }" % p0
    if kind == "random_access":
        return "RandomAccessGate { bits: %d, num_copies: %d, num_extra_constants: %d, _phantom: PhantomData<%s> }<D=2>" % (p0, p1, p2, FIELD)
    if kind == "exponentiation":
        return "ExponentiationGate { num_power_bits: %d, _phantom: PhantomData<%s> }<D=2>" % (p0, FIELD)
    if kind == "coset_interpolation":
        w = ", ".join(str(x) for x in barycentric_weights(p0)[1])
        return "CosetInterpolationGate { subgroup_bits: %d, degree: %d, barycentric_weights: [%s], _phantom: PhantomData<%s> }<D=2>" % (p0, p1, w, FIELD)
    raise ValueError(kind)


class Gate:
    def __init__(self, kind, p0=0, p1=0, p2=0):
        self.kind = kind
        self.p0, self.p1, self.p2 = default_params(kind, p0, p1, p2)
        self.degree, self.num_constraints, self.num_constants, self.num_wires = derived(kind, self.p0, self.p1, self.p2)
        self.id = gate_id(kind, self.p0, self.p1, self.p2)
        self.selector_index = self.group_start = self.group_end = self.index = 0


class GateSet:
    """gates sorted by (degree, id) and grouped as gates/selectors.rs selector_polynomials does"""

    def __init__(self, spec, max_degree=SELECTOR_MAX_DEGREE):
        gs = [Gate(*((s,) if isinstance(s, str) else s)) for s in spec]
        gs.sort(key=lambda g: (g.degree, g.id))
        for i, g in enumerate(gs):
            g.index = i
        n = len(gs)
        if gs[-1].degree + n - 1 <= max_degree:
            groups = [(0, n)]
        else:
            assert gs[-1].degree < max_degree
            groups, start = [], 0
            while start < n:
                size = 0
                while start + size < n and size + gs[start + size].degree < max_degree:
                    size += 1
                groups.append((start, start + size))
                start += size
        for si, (a, b) in enumerate(groups):
            for g in gs[a:b]:
                g.selector_index, g.group_start, g.group_end = si, a, b
        self.gates, self.num_selectors = gs, len(groups)
        self.num_gate_constraints = max(g.num_constraints for g in gs)
        self.num_constants = max(g.num_constants for g in gs)
        self.arr = (OrcGate * n)()
        for o, g in zip(self.arr, gs):
            o.kind, o.p0, o.p1, o.p2 = KINDS.index(g.kind), g.p0, g.p1, g.p2
            o.selector_index, o.group_start, o.group_end, o.index = g.selector_index, g.group_start, g.group_end, g.index

    def by_kind(self, kind):
        return next(g for g in self.gates if g.kind == kind)

    def c_gate(self, gate):
        return self.arr[gate.index]

    def selector_values(self, gate):
        return [gate.index if s == gate.selector_index else UNUSED_SELECTOR for s in range(self.num_selectors)]

    # ---- oracle/gates.c ----
    def terms_coset(self, constants_coeffs, wires_coeffs, pi_hash, alphas):
        """folded gate terms on the coset in natural order: [nc][8n]"""
        c, w = orc.u64arr(constants_coeffs), orc.u64arr(wires_coeffs)
        n = w.shape[1]
        a, h = orc.u64arr(alphas), orc.u64arr(pi_hash)
        out = np.zeros((a.size, 8 * n), np.uint64)
        rc = _bind().orc_gate_terms_coset(self.arr, len(self.gates), self.num_selectors, orc.ptr(c), c.shape[0], orc.ptr(w), w.shape[0],
                                          n.bit_length() - 1, orc.ptr(h), orc.ptr(a), a.size, orc.ptr(out))
        assert rc == 0, rc
        return out

    def terms_zeta(self, constants_z, wires_z, pi_hash, alphas):
        c, w, a, h = orc.u64arr(constants_z), orc.u64arr(wires_z), orc.u64arr(alphas), orc.u64arr(pi_hash)
        out = np.zeros((a.size, 2), np.uint64)
        rc = _bind().orc_gate_terms_zeta(self.arr, len(self.gates), self.num_selectors, orc.ptr(c), c.shape[0], orc.ptr(w), w.shape[0],
                                         orc.ptr(h), orc.ptr(a), a.size, orc.ptr(out))
        assert rc == 0, rc
        return out

    def eval_row(self, gate, row, constants, pi_hash=(0, 0, 0, 0)):
        """constraint values (0 = satisfied) of one gate on one base-field trace row"""
        r, c, h = orc.u64arr(row), orc.u64arr(list(constants) + [0]), orc.u64arr(pi_hash)
        out = np.zeros(256, np.uint64)
        k = _bind().orc_gate_eval_row(C.byref(self.arr[gate.index]), orc.ptr(r), r.size, orc.ptr(c), len(constants), orc.ptr(h), orc.ptr(out))
        assert k == gate.num_constraints, (k, gate.num_constraints)
        return out[:k]


# ---------------- pure-Python witness generators (iop/generator.rs SimpleGenerator::run_once of every gate) ----------------
def _amul(x, y):  # (a + bX)(c + dX), X^2 = 7
    return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def _aadd(x, y):
    return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)


def _ascale(x, s):
    return (x[0] * s % P, x[1] * s % P)


def _mds(s):
    C_ = pymodel.CIRC
    return [(sum(s[(i + r) % 12] * C_[i] for i in range(12)) + (8 * s[0] if r == 0 else 0)) % P for r in range(12)]


def witness_row(gate, rng, constants=(), pi_hash=(0, 0, 0, 0), num_wires=NUM_WIRES, preset=None):
    """A trace row satisfying `gate`: free inputs drawn from rng (a random.Random) or taken from preset {wire: value}, the
    rest generated.  Unused wires random."""
    rnd = lambda: rng.randrange(P)
    row = [rnd() for _ in range(num_wires)]
    for i, v in (preset or {}).items():
        row[i] = int(v) % P
    k, p0, p1, p2 = gate.kind, gate.p0, gate.p1, gate.p2
    c = list(constants)
    if k == "constant":
        row[:p0] = c[:p0]
    elif k == "public_input":
        row[:4] = [int(x) for x in pi_hash]
    elif k == "arithmetic":
        for i in range(p0):
            row[4 * i + 3] = (row[4 * i] * row[4 * i + 1] * c[0] + row[4 * i + 2] * c[1]) % P
    elif k == "base_sum":
        x = rng.randrange(min(P, p1 ** p0))
        row[0] = x
        for i in range(p0):
            row[1 + i], x = x % p1, x // p1
    elif k == "poseidon":
        row[24] = rng.randrange(2) if not (preset and 24 in preset) else row[24]
        rc = pymodel.round_constants()
        st = row[:12]
        if row[24]:
            st[0:4], st[4:8] = st[4:8], st[0:4]
        for i in range(4):
            row[25 + i] = row[24] * (row[i + 4] - row[i]) % P
        for r in range(30):
            st = [(st[i] + rc[12 * r + i]) % P for i in range(12)]
            if r < 4 or r >= 26:
                if 1 <= r < 4:
                    row[29 + 12 * (r - 1):29 + 12 * r] = st
                if r >= 26:
                    row[87 + 12 * (r - 26):87 + 12 * (r - 25)] = st
                st = [pow(x, 7, P) for x in st]
            else:
                row[65 + r - 4] = st[0]
                st[0] = pow(st[0], 7, P)
            st = _mds(st)
        row[12:24] = st
    elif k == "poseidon_mds":
        ins = [(row[2 * i], row[2 * i + 1]) for i in range(12)]
        for comp in range(2):
            out = _mds([x[comp] for x in ins])
            for r in range(12):
                row[24 + 2 * r + comp] = out[r]
    elif k == "arithmetic_ext":
        for i in range(p0):
            w = lambda j: (row[8 * i + 2 * j], row[8 * i + 2 * j + 1])
            o = _aadd(_ascale(_amul(w(0), w(1)), c[0]), _ascale(w(2), c[1]))
            row[8 * i + 6], row[8 * i + 7] = o
    elif k == "mul_ext":
        for i in range(p0):
            o = _ascale(_amul((row[6 * i], row[6 * i + 1]), (row[6 * i + 2], row[6 * i + 3])), c[0])
            row[6 * i + 4], row[6 * i + 5] = o
    elif k in ("reducing", "reducing_ext"):
        ext = k == "reducing_ext"
        n = p0
        start_accs = 6 + 2 * n if ext else 6 + n
        alpha, acc = (row[2], row[3]), (row[4], row[5])
        for i in range(n):
            coeff = (row[6 + 2 * i], row[6 + 2 * i + 1]) if ext else (row[6 + i], 0)
            acc = _aadd(_amul(acc, alpha), coeff)
            at = 0 if i == n - 1 else start_accs + 2 * i
            row[at], row[at + 1] = acc
    elif k == "random_access":
        bits, copies, extra, vec = p0, p1, p2, 1 << p0
        routed = (2 + vec) * copies + extra
        for cp in range(copies):
            base = (2 + vec) * cp
            idx = rng.randrange(vec)
            row[base] = idx
            row[base + 1] = row[base + 2 + idx]
            for b in range(bits):
                row[routed + cp * bits + b] = (idx >> b) & 1
        for i in range(extra):
            row[(2 + vec) * copies + i] = c[i]
    elif k == "exponentiation":
        n = p0
        e = rng.getrandbits(n)
        for i in range(n):
            row[1 + i] = (e >> i) & 1
        prev = 1
        for i in range(n):
            sq = 1 if i == 0 else prev * prev % P
            prev = sq * row[0] % P if row[1 + (n - 1 - i)] else sq
            row[2 + n + i] = prev
        row[1 + n] = prev
        assert prev == pow(row[0], e, P)
    elif k == "coset_interpolation":
        bits, degree = p0, p1
        points = 1 << bits
        ni = (points - 2) // (degree - 1)
        s_point, s_value = 1 + 2 * points, 1 + 2 * points + 2
        s_inter = s_value + 2
        s_shifted = s_inter + 4 * ni
        dom, wts = barycentric_weights(bits)
        shift = row[0] or 1
        row[0] = shift
        sinv = pow(shift, P - 2, P)
        shifted = _ascale((row[s_point], row[s_point + 1]), sinv)
        row[s_shifted], row[s_shifted + 1] = shifted
        ev, pr = (0, 0), (1, 0)
        bounds = [(0, min(degree, points))] + [(1 + (degree - 1) * (i + 1), min(1 + (degree - 1) * (i + 2), points)) for i in range(ni)]
        for ci, (a, b) in enumerate(bounds):
            if ci > 0:
                row[s_inter + 2 * (ci - 1)], row[s_inter + 2 * (ci - 1) + 1] = ev
                row[s_inter + 2 * (ni + ci - 1)], row[s_inter + 2 * (ni + ci - 1) + 1] = pr
            for i in range(a, b):
                term = ((shifted[0] - dom[i]) % P, shifted[1])
                val = _ascale((row[1 + 2 * i], row[2 + 2 * i]), wts[i])
                ev = _aadd(_amul(ev, term), _amul(val, pr))
                pr = _amul(pr, term)
        row[s_value], row[s_value + 1] = ev
    return row


def interpolate_check(row, gate):
    """independent meaning check of a coset-interpolation row: evaluation_value == P(evaluation_point) where P interpolates the
    values on the coset shift * <g> (Lagrange, big-int)"""
    bits = gate.p0
    points = 1 << bits
    dom, _ = barycentric_weights(bits)
    shift = row[0]
    xs = [shift * d % P for d in dom]
    s_point, s_value = 1 + 2 * points, 1 + 2 * points + 2
    z = (row[s_point], row[s_point + 1])
    total = (0, 0)
    for i in range(points):
        num, den = (1, 0), 1
        for j in range(points):
            if j != i:
                num = _amul(num, ((z[0] - xs[j]) % P, z[1]))
                den = den * (xs[i] - xs[j]) % P
        li = _ascale(num, pow(den, P - 2, P))
        total = _aadd(total, _amul(li, (row[1 + 2 * i], row[2 + 2 * i])))
    return total == (row[s_value], row[s_value + 1])


def demo_circuit(rng, gs, log_n, public_inputs, n_routed=NUM_ROUTED, describe=False):
    """A small but real circuit over the gate set gs (must contain public_input, poseidon, arithmetic, noop; any other gate of
    gs gets a few unconnected rows):
      row 0            PublicInputGate: wires 0..4 = hash of the public inputs
      row 1            PoseidonGate hashing (pi_0..pi_3, 0^8): outputs 0..4 copy-constrained to row 0  (hash_n_to_hash_no_pad in-circuit)
      rows 2..5        a Poseidon chain: inputs of row r = outputs of row r-1 (12 copy constraints each)
      row 6            ArithmeticGate fed by copies of chain outputs, ops chained output -> next multiplicand
      then             two rows of every other gate type with free inputs; the rest NoopGate
    Returns constants [num_selectors + num_constants][n], wires [135][n], sigma values [n_routed][n], pi_hash."""
    n = 1 << log_n
    assert len(public_inputs) == 4
    pi_hash = pymodel.hash_no_pad(public_inputs)
    n_const = gs.num_selectors + gs.num_constants
    constants = np.zeros((n_const, n), np.uint64)
    wires = np.zeros((NUM_WIRES, n), np.uint64)
    classes = []  # lists of (col, row) that must carry equal values

    def put(r, gate, preset=None, consts=None):
        consts = [rng.randrange(P) for _ in range(gate.num_constants)] if consts is None else consts
        constants[:gs.num_selectors, r] = gs.selector_values(gate)
        constants[gs.num_selectors:gs.num_selectors + len(consts), r] = consts
        row = witness_row(gate, rng, consts, pi_hash, preset=preset)
        wires[:, r] = row
        return row

    pos, ar, pub, noop = gs.by_kind("poseidon"), gs.by_kind("arithmetic"), gs.by_kind("public_input"), gs.by_kind("noop")
    put(0, pub)
    prev = put(1, pos, {**{i: public_inputs[i] for i in range(4)}, **{i: 0 for i in range(4, 12)}, 24: 0})
    assert prev[12:16] == pi_hash
    classes += [[(i, 0), (12 + i, 1)] for i in range(4)]
    for r in range(2, 6):
        cur = put(r, pos, {**{i: prev[12 + i] for i in range(12)}, 24: r & 1})
        classes += [[(12 + i, r - 1), (i, r)] for i in range(12)]
        prev = cur
    c01 = [rng.randrange(P), rng.randrange(P)]
    pre = {0: prev[12], 1: prev[13], 2: prev[14]}
    classes += [[(12, 5), (0, 6)], [(13, 5), (1, 6)], [(14, 5), (2, 6)]]
    # chain the ops: multiplicand_0 of op i+1 = output of op i (computed on the fly)
    out = (pre[0] * pre[1] * c01[0] + pre[2] * c01[1]) % P
    vals = [rng.randrange(P) for _ in range(4 * ar.p0)]
    vals[0:3] = [pre[0], pre[1], pre[2]]
    for i in range(1, ar.p0):
        vals[4 * i] = out
        classes.append([(4 * (i - 1) + 3, 6), (4 * i, 6)])
        out = (vals[4 * i] * vals[4 * i + 1] * c01[0] + vals[4 * i + 2] * c01[1]) % P
    put(6, ar, {4 * i + j: vals[4 * i + j] for i in range(ar.p0) for j in range(3)}, c01)
    r = 7
    for g in gs.gates:
        if g.kind in ("poseidon", "arithmetic", "public_input", "noop"):
            continue
        for _ in range(2):
            assert r < n
            put(r, g)
            r += 1
    if r < n:  # NoopGate rows: unconstrained wires, filled in bulk (the full-size test has 2^15 of them)
        seed = rng.getrandbits(63)
        fill = np.random.default_rng(seed).integers(0, P, size=(NUM_WIRES, n - r), dtype=np.uint64)
        wires[:, r:] = fill
        sel = gs.selector_values(noop)
        for s_i, v in enumerate(sel):
            constants[s_i, r:] = v
    # sigma: identity except one cycle per class (plonk/permutation_argument.rs: sigma maps a position to the next in its class)
    w = pymodel.root_of_unity(log_n)
    wp = [pow(w, i, P) for i in range(n)]
    kp = [pow(7, c, P) for c in range(n_routed)]
    wp_arr = np.array(wp, dtype=object)
    sigma = np.stack([np.array((wp_arr * kp[c]) % P, dtype=np.uint64) for c in range(n_routed)])
    # merge overlapping pairs into equivalence classes (the copy-constraint forest of CircuitBuilder), one cycle per class
    parent = {}

    def find(x):
        parent.setdefault(x, x)
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for cl in classes:
        for a, b in zip(cl, cl[1:]):
            parent[find(a)] = find(b)
    merged = {}
    for x in list(parent):
        merged.setdefault(find(x), []).append(x)
    for cl in merged.values():
        cl.sort(key=lambda x: (x[1], x[0]))   # WirePartition order: by (row, column)
        assert all(c < n_routed for c, _ in cl) and len({int(wires[c, rr]) for c, rr in cl}) == 1
        for (c, rr), (c2, r2) in zip(cl, cl[1:] + cl[:1]):
            sigma[c, rr] = kp[c2] * wp[r2] % P
    if describe:
        # the circuit as data (what CircuitBuilder::build leaves behind) + the PartialWitness a caller would set: every wire
        # that no gate generator owns and that is not reachable through a copy constraint from a generated wire
        row_gate = np.zeros(n, np.uint32)
        for rr in range(n):
            sel = [int(x) for x in constants[:gs.num_selectors, rr]]
            row_gate[rr] = next(g.index for g in gs.gates if gs.selector_values(g) == sel)
        pairs = [(a[0] * n + a[1], b[0] * n + b[1]) for cl in classes for a, b in zip(cl, cl[1:])]
        return constants, wires, sigma, pi_hash, {"row_gate": row_gate, "copies": pairs, "classes": [sorted(c) for c in merged.values()]}
    return constants, wires, sigma, pi_hash
