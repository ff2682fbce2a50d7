"""CPU tests (-m "not gpu") of the gate constraints: the C oracle (oracle/gates.c), the independent Python witness
generators, and the product's HOST-side gate code (layout, evaluation at an extension point, witness rows) --
no device compute is called."""
import random

import numpy as np
import pytest

import gates_oracle as go
import oracle as orc
import pymodel
from vpbs_amd import api

P = orc.P
ALL = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing",
       "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"]
VARIANTS = [("base_sum", 10, 3), ("base_sum", 20, 4), ("random_access", 1), ("random_access", 2), ("random_access", 3), ("random_access", 5),
            ("coset_interpolation", 2), ("coset_interpolation", 3), ("coset_interpolation", 5), ("coset_interpolation", 4, 3),
            ("arithmetic", 3), ("constant", 1), ("reducing", 5), ("reducing_ext", 1), ("exponentiation", 7), ("mul_ext", 2),
            ("arithmetic_ext", 1)]


def _spec_name(s):
    return s if isinstance(s, str) else s[0]


def test_standard_config_parameters():
    """the *_from_config numbers under standard_recursion_config (135 wires, 80 routed, 2 constants)"""
    g = {(_spec_name(s)): go.Gate(*((s,) if isinstance(s, str) else s)) for s in ALL}
    assert g["arithmetic"].p0 == 20 and g["arithmetic_ext"].p0 == 10 and g["mul_ext"].p0 == 13
    assert (g["base_sum"].p0, g["base_sum"].p1) == (63, 2)
    assert g["reducing"].p0 == 43 and g["reducing_ext"].p0 == 32 and g["exponentiation"].p0 == 66
    assert (g["random_access"].p0, g["random_access"].p1, g["random_access"].p2) == (4, 4, 2)
    assert (g["coset_interpolation"].p0, g["coset_interpolation"].p1) == (4, 6)
    assert g["poseidon"].num_constraints == 123 and g["poseidon"].num_wires == 135
    for x in g.values():
        assert x.num_wires <= 135 and x.degree <= 7


def test_product_layout_matches_python_restatement():
    for spec in (ALL, ["noop", "arithmetic", "public_input"], ["poseidon", "constant"], ALL[:8] + VARIANTS[:3]):
        a, b = go.GateSet(spec), api.GateSet(spec)
        assert (a.num_selectors, a.num_gate_constraints, a.num_constants) == (b.num_selectors, b.num_gate_constraints, b.num_constants)
        assert [g.id for g in a.gates] == b.ids()
        for x, y in zip(a.gates, b):
            assert (go.KINDS[y.kind], y.p0, y.p1, y.p2) == (x.kind, x.p0, x.p1, x.p2)
            assert (y.degree, y.num_constraints, y.num_constants, y.num_wires) == (x.degree, x.num_constraints, x.num_constants, x.num_wires)
            assert (y.selector_index, y.group_start, y.group_end, y.index) == (x.selector_index, x.group_start, x.group_end, x.index)
    full = go.GateSet(ALL)
    assert full.gates[-1].kind == "poseidon" and full.gates[0].kind == "noop"
    # every group obeys the degree bound the quotient relies on: filter degree (|G| factors) + gate degree <= 9 = quotient_degree_factor + 1
    for g in full.gates:
        assert (g.group_end - g.group_start) + g.degree <= 9


@pytest.mark.parametrize("spec", ALL + VARIANTS, ids=lambda s: s if isinstance(s, str) else "-".join(map(str, s)))
def test_generated_rows_satisfy_oracle_and_product(spec):
    """Python generator row -> zero constraints in oracle/gates.c and in the product's host evaluation; a corrupted wire is
    caught; the product's own generator (vpbs_gate_fill_row) reproduces the Python row."""
    rng = random.Random(hash(str(spec)) & 0xFFFF)
    gs, ps = go.GateSet([spec, "noop"] if spec != "noop" else ["noop", "constant"]), None
    ps = api.GateSet([spec, "noop"] if spec != "noop" else ["noop", "constant"])
    name = _spec_name(spec)
    gate, pgate = gs.by_kind(name), ps.by_kind(name)
    pi_hash = [rng.randrange(P) for _ in range(4)]
    alphas = [rng.randrange(P) for _ in range(2)]
    for trial in range(3):
        consts = [rng.randrange(P) for _ in range(gate.num_constants)]
        row = go.witness_row(gate, rng, consts, pi_hash)
        vals = gs.eval_row(gate, row, consts, pi_hash)
        assert not vals.any(), "oracle: generated row violates constraint %s" % np.nonzero(vals)[0]
        # product host evaluation at a "point" whose openings are the base-field row values: folded sum must be zero
        sel = ps.selector_values(pgate)
        c_at = np.array([[v, 0] for v in sel + consts + [0] * (ps.num_constants - len(consts))], np.uint64)
        w_at = np.array([[v, 0] for v in row], np.uint64)
        assert not ps.terms_at(c_at, w_at, pi_hash, alphas).any()
        # the product's generator fills the same row from the same free inputs
        if name not in ("public_input", "noop"):
            mine = np.array(row, np.uint64)
            keep = mine.copy()
            # scramble every wire the generator owns, then regenerate
            owned = _owned_wires(gate)
            for i in owned:
                mine[i] = rng.randrange(P)
            api.GateSet.fill_row(pgate, consts, mine)
            assert (mine == keep).all(), np.nonzero(mine != keep)[0]
        # a single corrupted wire breaks at least one constraint (both implementations)
        if gate.num_constraints:
            i = rng.choice(_owned_wires(gate) or list(range(min(4, gate.num_wires))))  # a wire some constraint pins down
            bad = list(row)
            bad[i] = (bad[i] + 1 + rng.randrange(P - 1)) % P
            assert gs.eval_row(gate, bad, consts, pi_hash).any()
            w_bad = np.array([[v, 0] for v in bad], np.uint64)
            assert ps.terms_at(c_at, w_bad, pi_hash, alphas).any()


def _owned_wires(gate):
    """wires written by the gate's generators (the rest are free inputs)"""
    k, p0, p1, p2 = gate.kind, gate.p0, gate.p1, gate.p2
    if k == "constant":
        return list(range(p0))
    if k == "arithmetic":
        return [4 * i + 3 for i in range(p0)]
    if k == "base_sum":
        return list(range(1, 1 + p0))
    if k == "poseidon":
        return list(range(12, 24)) + list(range(25, 135))
    if k == "poseidon_mds":
        return list(range(24, 48))
    if k == "arithmetic_ext":
        return [8 * i + 6 + j for i in range(p0) for j in range(2)]
    if k == "mul_ext":
        return [6 * i + 4 + j for i in range(p0) for j in range(2)]
    if k == "reducing":
        return [0, 1] + list(range(6 + p0, 6 + p0 + 2 * (p0 - 1)))
    if k == "reducing_ext":
        return [0, 1] + list(range(6 + 2 * p0, 6 + 2 * p0 + 2 * (p0 - 1)))
    if k == "random_access":
        vec = 1 << p0
        routed = (2 + vec) * p1 + p2
        return [(2 + vec) * c + 1 for c in range(p1)] + list(range((2 + vec) * p1, routed + p1 * p0))
    if k == "exponentiation":
        return list(range(1 + p0, 2 + 2 * p0))
    if k == "coset_interpolation":
        points = 1 << p0
        return list(range(1 + 2 * points + 2, gate.num_wires))
    return []


def test_poseidon_gate_row_is_the_pinned_permutation():
    """the PoseidonGate witness (checked against the constraints above) carries the KAT-pinned permutation on its output wires"""
    rng = random.Random(5)
    gs = go.GateSet(["poseidon", "noop"])
    gate = gs.by_kind("poseidon")
    for _ in range(3):
        row = go.witness_row(gate, rng)
        ins = row[:12]
        if row[24]:
            ins = ins[4:8] + ins[0:4] + ins[8:12]
        assert row[12:24] == [int(x) for x in orc.poseidon(ins)] == pymodel.poseidon(ins)


def test_coset_interpolation_row_means_lagrange_interpolation():
    rng = random.Random(9)
    for spec in (("coset_interpolation", 2), ("coset_interpolation", 4), ("coset_interpolation", 3, 3)):
        gs = go.GateSet([spec, "noop"])
        gate = gs.by_kind("coset_interpolation")
        row = go.witness_row(gate, rng)
        assert not gs.eval_row(gate, row, []).any()
        assert go.interpolate_check(row, gate)


def test_gate_semantics_spot_checks():
    """meaning of a few generated rows, beyond 'constraints vanish'"""
    rng = random.Random(3)
    gs = go.GateSet(["base_sum", "exponentiation", ("random_access", 3), "reducing", "noop"])
    g = gs.by_kind("base_sum")
    row = go.witness_row(g, rng)
    assert sum(b << i for i, b in enumerate(row[1:1 + g.p0])) == row[0] and set(row[1:1 + g.p0]) <= {0, 1}
    g = gs.by_kind("exponentiation")
    row = go.witness_row(g, rng)
    e = sum(b << i for i, b in enumerate(row[1:1 + g.p0]))
    assert row[1 + g.p0] == pow(row[0], e, P)
    g = gs.by_kind("random_access")
    row = go.witness_row(g, rng, [1, 2])
    for c in range(g.p1):
        base = 10 * c
        assert row[base + 1] == row[base + 2 + row[base]]
    g = gs.by_kind("reducing")
    row = go.witness_row(g, rng)
    # output = old_acc * alpha^n + sum coeff_i alpha^(n-1-i) in GF(p^2)
    alpha, acc = (row[2], row[3]), (row[4], row[5])
    for i in range(g.p0):
        acc = pymodel.ext_mul(acc, alpha)
        acc = ((acc[0] + row[6 + i]) % P, acc[1])
    assert acc == (row[0], row[1])


def _random_circuit(rng, gs, ps, log_n, kinds_cycle):
    """a trace whose row r holds gate kinds_cycle[r % len]: returns constants [cols][n], wires [135][n] (values on H)"""
    n = 1 << log_n
    n_const = gs.num_selectors + gs.num_constants
    constants = np.zeros((n_const, n), np.uint64)
    wires = np.zeros((135, n), np.uint64)
    pi_hash = [rng.randrange(P) for _ in range(4)]
    for r in range(n):
        gate = gs.by_kind(kinds_cycle[r % len(kinds_cycle)])
        consts = [rng.randrange(P) for _ in range(gate.num_constants)]
        constants[:gs.num_selectors, r] = gs.selector_values(gate)
        constants[gs.num_selectors:gs.num_selectors + len(consts), r] = consts
        wires[:, r] = go.witness_row(gate, rng, consts, pi_hash)
    return constants, wires, pi_hash


def test_folded_terms_vanish_on_the_subgroup_and_are_divisible_by_zh():
    """A satisfying trace: the folded gate terms vanish on H, i.e. terms = Z_H * q with deg q < 8n (filter * constraint has degree
    <= 9(n-1)).  q is interpolated from the coset values of terms / Z_H and the identity is checked at a random extension point
    against the verifier-side evaluation (both implementations); a broken row destroys it."""
    rng = random.Random(11)
    spec = ["noop", "constant", "public_input", "arithmetic", ("base_sum", 20, 2), "poseidon", "mul_ext", ("random_access", 2)]
    gs, ps = go.GateSet(spec), api.GateSet(spec)
    log_n = 4
    n = 1 << log_n
    big = 8 * n
    kinds = [g.kind for g in gs.gates]
    constants, wires, pi_hash = _random_circuit(rng, gs, ps, log_n, kinds)
    alphas = [rng.randrange(P) for _ in range(2)]
    c_coeffs = np.stack([orc.fft(c, inverse=True) for c in constants])
    zeta = (rng.randrange(P), rng.randrange(P))
    zeta_n = zeta
    for _ in range(log_n):
        zeta_n = pymodel.ext_mul(zeta_n, zeta_n)
    zh_zeta = ((zeta_n[0] - 1) % P, zeta_n[1])
    w8n = pymodel.root_of_unity(log_n + 3)
    inv7 = pow(7, P - 2, P)
    for broken in (False, True):
        w = wires.copy()
        if broken:
            row = 3
            gate = gs.by_kind(kinds[row % len(kinds)])
            col = (_owned_wires(gate) or [0])[0]
            w[col, row] = (int(w[col, row]) + 1) % P
        w_coeffs = np.stack([orc.fft(c, inverse=True) for c in w])
        terms = gs.terms_coset(c_coeffs, w_coeffs, pi_hash, alphas)  # [2][8n] natural coset order
        cz, wz = orc.eval_coeffs_ext(c_coeffs, zeta), orc.eval_coeffs_ext(w_coeffs, zeta)
        tz_orc = gs.terms_zeta(cz, wz, pi_hash, alphas)
        assert (tz_orc == ps.terms_at(cz, wz, pi_hash, alphas)).all()
        for a in range(2):
            q = [int(terms[a][t]) * pow(pow(7 * pow(w8n, t, P) % P, n, P) - 1, P - 2, P) % P for t in range(big)]
            coeffs = orc.fft(np.array(q, np.uint64), inverse=True)  # coefficients of q(7 X)
            qc = np.array([int(c) * pow(inv7, i, P) % P for i, c in enumerate(coeffs)], np.uint64)
            qz = tuple(int(v) for v in orc.eval_coeffs_ext(qc[None, :], zeta)[0])
            holds = pymodel.ext_mul(qz, zh_zeta) == (int(tz_orc[a][0]), int(tz_orc[a][1]))
            assert holds == (not broken)


def test_product_terms_at_matches_oracle_on_random_openings():
    """bit-exact agreement of the two implementations at random extension points with random (unsatisfying) wires"""
    rng = random.Random(21)
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    n_const = gs.num_selectors + gs.num_constants
    for _ in range(4):
        cz = np.array([[rng.randrange(P), rng.randrange(P)] for _ in range(n_const)], np.uint64)
        wz = np.array([[rng.randrange(P), rng.randrange(P)] for _ in range(135)], np.uint64)
        pi_hash = [rng.randrange(P) for _ in range(4)]
        alphas = [rng.randrange(P) for _ in range(3)]
        assert (gs.terms_zeta(cz, wz, pi_hash, alphas) == ps.terms_at(cz, wz, pi_hash, alphas)).all()


def test_gate_argument_errors():
    with pytest.raises(api.VpbsError):
        api.GateSet([("random_access", 6)])
    with pytest.raises(api.VpbsError):
        api.GateSet([("base_sum", 4, 1)])
    with pytest.raises(api.VpbsError):
        api.GateSet(["arithmetic", "arithmetic"])  # a gate type appears once in a gate set
    ps = api.GateSet(["arithmetic", "noop"])
    with pytest.raises(api.VpbsError):
        ps.terms_at(np.zeros((1, 2), np.uint64), np.zeros((135, 2), np.uint64), [0] * 4, [1])  # constants columns missing
    g = ps.by_kind("arithmetic")
    row = np.zeros(10, np.uint64)
    bs = api.GateSet([("base_sum", 4, 2), "noop"]).by_kind("base_sum")
    row[0] = 16  # does not fit 4 bits
    with pytest.raises(api.VpbsError):
        api.GateSet.fill_row(bs, None, row)


DIGEST = [11, 22, 33, 44]


def _check_oracle(gs, proof, ncols, n_constants, n_routed, pi_hash, log_n):
    """oracle verifier: FRI + vanishing(zeta) == Z_H(zeta) t(zeta) with the gate terms evaluated from the openings"""
    op = proof["openings"]
    n_cs, n_w, n_z, n_q = ncols
    cs_z, w_z = op[:n_cs], op[n_cs:n_cs + n_w]
    zs_all, q_z, zs_next = op[n_cs + n_w:n_cs + n_w + n_z], op[n_cs + n_w + n_z:n_cs + n_w + n_z + n_q], op[n_cs + n_w + n_z + n_q:]
    ch = [int(x) for x in proof["challenges"]]
    betas, gammas, alphas, zeta = ch[0:2], ch[2:4], ch[4:6], ch[6:8]
    gt = gs.terms_zeta(cs_z[:n_constants], w_z, pi_hash, alphas)
    return orc.check_vanishing_at_zeta(w_z[:n_routed], cs_z[n_constants:], zs_all[:2], zs_next, zs_all[2:], q_z, log_n, betas, gammas, alphas,
                                       zeta, gate_terms_zeta=gt)


def test_demo_circuit_oracle_proof_verifies_under_both_verifiers():
    """CPU only: the oracle proves a circuit with Poseidon / arithmetic / public-input gates, copy constraints and a few rows of
    every other gate; the oracle's verifier and the PRODUCT's host verifier (gate constraints evaluated at zeta from the
    openings) both accept; one wrong witness value makes both reject the vanishing identity."""
    import step_oracle
    rng = random.Random(77)
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    log_n, n_routed = 6, 80
    pis = [rng.randrange(P) for _ in range(4)]
    constants, wires, sigma, pi_hash = go.demo_circuit(rng, gs, log_n, pis)
    n_constants = constants.shape[0]
    ncols = [n_constants + n_routed, 135, 20, 16]
    for broken in (False, True):
        w = wires.copy()
        if broken:
            w[14, 3] = (int(w[14, 3]) + 1) % P  # an output wire of a Poseidon row
        inputs = {"constants_sigmas": np.concatenate([constants, sigma]), "wires": w, "quotient": None}
        proof = step_oracle.prove_step(inputs, DIGEST, pis, log_n, sigmas=sigma, n_routed=n_routed, n_constants=n_constants, gates=gs)
        assert step_oracle.verify_step(proof, proof["cs_cap"], ncols, DIGEST, pis, log_n)   # FRI: commitments are consistent either way
        assert _check_oracle(gs, proof, ncols, n_constants, n_routed, pi_hash, log_n) == (not broken)
        assert api.verify_step(proof, proof["cs_cap"], ncols, DIGEST, pis, log_n, check_permutation=True, n_constants=n_constants,
                               n_routed=n_routed, gates=ps) == (not broken)
        if not broken:
            # without the gate constraints the identity must fail (the quotient contains them), and wrong public inputs too
            assert not api.verify_step(proof, proof["cs_cap"], ncols, DIGEST, pis, log_n, check_permutation=True,
                                       n_constants=n_constants, n_routed=n_routed)
            assert not api.verify_step(proof, proof["cs_cap"], ncols, DIGEST, [pis[0] ^ 1] + pis[1:], log_n, check_permutation=True,
                                       n_constants=n_constants, n_routed=n_routed, gates=ps)


def _free_inputs(gate):
    """wires a caller must provide for a stand-alone gate row: everything the gate uses minus what its generators own"""
    owned = set(_owned_wires(gate))
    return [w for w in range(gate.num_wires) if w not in owned]


def test_witness_generation_reproduces_the_demo_circuit():
    """Product host code (vpbs_generate_witness, vpbs_sigma_values, vpbs_selector_columns) against the independent Python build of
    the same circuit: from the PartialWitness alone (public inputs, free gate inputs) the scheduler regenerates every generated
    wire -- through the copy constraints: Poseidon chain, arithmetic chain, in-circuit public-input hash -- and the trace
    satisfies every gate and every copy constraint."""
    rng = random.Random(31)
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    log_n = 6
    n = 1 << log_n
    pis = [rng.randrange(P) for _ in range(4)]
    constants, wires, sigma, pi_hash, desc = go.demo_circuit(rng, gs, log_n, pis, describe=True)
    circ = api.Circuit(ps, log_n, desc["row_gate"], constants, desc["copies"])
    assert (circ.selector_columns() == constants[:gs.num_selectors]).all()
    assert (circ.sigma_values() == sigma).all()
    # PartialWitness: free inputs of every gate row that are not fed by a copy constraint from a generated wire
    generated = set()
    for r in range(n):
        g = gs.gates[int(desc["row_gate"][r])]
        generated |= {(w, r) for w in _owned_wires(g)}
    fed = set()
    for cl in desc["classes"]:
        if any(tuple(x) in generated for x in cl):
            fed |= {tuple(x) for x in cl}
    presets = {}
    for r in range(n):
        g = gs.gates[int(desc["row_gate"][r])]
        if g.kind == "public_input":
            continue   # its wires are copy-constrained to the in-circuit hash
        for w in _free_inputs(g):
            if (w, r) not in fed:
                presets[(w, r)] = int(wires[w, r])
    got = circ.generate_witness(presets)
    for r in range(n):
        g = gs.gates[int(desc["row_gate"][r])]
        used = range(g.num_wires)
        assert [int(got[w, r]) for w in used] == [int(wires[w, r]) for w in used], (r, g.kind)
        consts = [int(x) for x in constants[gs.num_selectors:gs.num_selectors + g.num_constants, r]]
        assert not gs.eval_row(g, got[:, r], consts, pi_hash).any()
    for cl in desc["classes"]:
        assert len({int(got[c, r]) for c, r in cl}) == 1
    assert [int(got[i, 0]) for i in range(4)] == pi_hash   # PublicInputGate wires = hash_no_pad(public inputs), via the circuit
    # the witness checker (integration aid): accepts the generated and the Python-built witness, names the first violation otherwise
    assert circ.check_witness(got, pi_hash) == (True, "")
    assert circ.check_witness(wires, pi_hash)[0]
    broken = got.copy()
    broken[3, 6] = (int(broken[3, 6]) + 1) % P                     # output of arithmetic op 0 on row 6
    ok, msg = circ.check_witness(broken, pi_hash)
    assert not ok and "row 6" in msg and "ArithmeticGate" in msg
    broken = got.copy()
    broken[0, 2] = (int(broken[0, 2]) + 1) % P                     # an input of the Poseidon chain: gate row 2 breaks first
    ok, msg = circ.check_witness(broken, pi_hash)
    assert not ok and "row 2" in msg
    broken = wires.copy()
    broken[:, 40] = 0                                              # a NoopGate row: nothing to violate ...
    assert circ.check_witness(broken, pi_hash)[0]
    ok, msg = circ.check_witness(got, [1, 2, 3, 4])                # ... but a wrong public-input hash breaks the PublicInputGate row
    assert not ok and "row 0" in msg and "PublicInputGate" in msg
    # errors: a conflicting preset, a missing input
    bad = dict(presets)
    bad[(12, 1)] = (int(wires[12, 1]) + 1) % P     # an output the Poseidon generator will set differently
    with pytest.raises(api.VpbsError, match="set twice"):
        circ.generate_witness(bad)
    missing = dict(presets)
    del missing[(0, 1)]
    with pytest.raises(api.VpbsError, match="weren't run"):
        circ.generate_witness(missing)


def test_gadget_level_generators_is_equal_split_le_le_sum():
    """The generator kinds the reference's own gadgets add outside the recursive verifier (builder.is_equal, cb.split_le, cb.le_sum:
    /root/reference/src/vtfhe/ivc_based_vpbs.rs:104-107 and the decomposition gadgets), scheduled together with the gate generators:
      is_equal(x, y):  EqualityGenerator -> (equal, inv); ArithmeticGate ops check diff * inv = 1 - equal and diff * equal = 0
      split_le(v, 64): WireSplitGenerator -> the sum wires of two BaseSumGate<2> rows (63 + 1 bits), BaseSplitGenerator -> limbs
      le_sum(bits):    BaseSumGenerator -> sum wire of a third BaseSumGate row fed with copies of the low bits
    The witness must satisfy every gate and copy constraint (vpbs_check_witness) and mean what the gadgets mean."""
    rng = random.Random(77)
    spec = ["noop", "arithmetic", "base_sum"]
    ps = api.GateSet(spec)
    ar, bs, noop = ps.by_kind("arithmetic"), ps.by_kind("base_sum"), ps.by_kind("noop")
    log_n = 3
    n = 1 << log_n
    for x, y, v in [(5, 5, 0xFFFFFFFF00000000), (rng.randrange(P), rng.randrange(P), rng.randrange(P)), (7, 8, 1), (0, 0, (1 << 63) + 12345)]:
        row_gate = np.full(n, noop.index, np.uint32)
        row_gate[0] = row_gate[1] = ar.index          # row 0: constants (1, -1): sub ; row 1: constants (1, 0): mul
        row_gate[2] = row_gate[3] = row_gate[4] = bs.index
        constants = np.zeros((ps.num_selectors + 2, n), np.uint64)
        constants[ps.num_selectors, 0], constants[ps.num_selectors + 1, 0] = 1, P - 1
        constants[ps.num_selectors, 1], constants[ps.num_selectors + 1, 1] = 1, 0
        pos = lambda c, r: (c, r)
        copies, presets, gens = [], {}, []
        cp = lambda a, b: copies.append((a[0] * n + a[1], b[0] * n + b[1]))
        # diff = x * 1 + (-1) * y           (row 0, op 0: wires 0..3)
        presets[(0, 0)], presets[(1, 0)], presets[(2, 0)] = x, 1, y
        # EqualityGenerator on (x, y) -> equal at (4, 1), inv at (1, 1)
        gens.append(("equality", 0, [pos(0, 0), pos(2, 0)], [pos(4, 1), pos(1, 1)]))
        # row 1 op 0: diff * inv            (wires 0..3: m0 = diff, m1 = inv, addend = 0)
        cp(pos(3, 0), pos(0, 1))
        presets[(2, 1)] = 0
        # row 1 op 1: diff * equal          (wires 4..7: m0 = equal, m1 = diff, addend = 0)
        cp(pos(3, 0), pos(5, 1))
        presets[(6, 1)] = 0
        # unused ops of rows 0, 1: free inputs
        for r in (0, 1):
            for op in range(2 if r else 1, ar.p0):
                for k in range(3):
                    presets[(4 * op + k, r)] = rng.randrange(P)
        # split_le(v, 64): rows 2 (bits 0..62) and 3 (bit 63)
        presets[(10, 5)] = v                                   # the integer lives on a noop row
        gens.append(("wire_split", 63, [pos(10, 5)], [pos(0, 2), pos(0, 3)]))
        # le_sum of the 20 low bits: row 4's limbs 0..19 are copies of row 2's, the others 0
        for i in range(bs.p0):
            if i < 20:
                cp(pos(1 + i, 2), pos(1 + i, 4))
            else:
                presets[(1 + i, 4)] = 0
        gens.append(("base_sum", 2, [pos(1 + i, 4) for i in range(bs.p0)], [pos(0, 4)]))
        circ = api.Circuit(ps, log_n, row_gate, constants, copies, generators=gens)
        constants[:ps.num_selectors] = circ.selector_columns()
        circ = api.Circuit(ps, log_n, row_gate, constants, copies, generators=gens)
        # row 3 would be generated twice for its sum wire: once by WireSplit (sum) and its limbs by the gate's BaseSplitGenerator
        # (gate generator watches the sum); row 4's BaseSplitGenerator also runs once le_sum has set the sum: consistent by construction
        w = circ.generate_witness(presets)
        ok, msg = circ.check_witness(w, [0, 0, 0, 0])
        assert ok, msg
        diff = (x - y) % P
        assert int(w[3, 0]) == diff and int(w[4, 1]) == (1 if x == y else 0)
        assert int(w[3, 1]) == (0 if x == y else 1)             # diff * inv
        assert int(w[7, 1]) == 0                                 # diff * equal
        bits = [int(w[1 + i, 2]) for i in range(63)] + [int(w[1, 3])]
        assert sum(b << i for i, b in enumerate(bits)) == v and all(int(w[1 + i, 3]) == 0 for i in range(1, 63))
        assert int(w[0, 4]) == v & ((1 << 20) - 1)
    # an integer that does not fit: split_le(v, 63) of a 64-bit value
    gens_bad = [("wire_split", 63, [(10, 5)], [(0, 2)])]
    row_gate = np.full(n, noop.index, np.uint32); row_gate[2] = bs.index
    circ = api.Circuit(ps, log_n, row_gate, np.zeros((ps.num_selectors + 2, n), np.uint64), [], generators=gens_bad)
    with pytest.raises(api.VpbsError, match="too large"):
        circ.generate_witness({(10, 5): 1 << 63})


def test_gadget_level_generators_of_the_recursive_verifier():
    """QuotientGeneratorExtension (div_extension: the quotient is a fresh target, an ArithmeticExtensionGate op checks den * q = num),
    LowHighGenerator (split_low_high: x = low + 2^n_log * high checked by an ArithmeticGate op) and CopyGenerator, scheduled with the
    gate generators; the checks close through copy constraints, so a wrong generator would set a class twice with different values."""
    rng = random.Random(99)
    spec = ["noop", "arithmetic", "arithmetic_ext"]
    ps = api.GateSet(spec)
    ar, ax, noop = ps.by_kind("arithmetic"), ps.by_kind("arithmetic_ext"), ps.by_kind("noop")
    log_n, n_log = 3, 20
    n = 1 << log_n
    for trial in range(4):
        num, den = [rng.randrange(P) for _ in range(2)], [rng.randrange(P), rng.randrange(P) if trial else 0]
        x = rng.randrange(P)
        row_gate = np.full(n, noop.index, np.uint32)
        row_gate[0], row_gate[1] = ax.index, ar.index
        constants = np.zeros((ps.num_selectors + 2, n), np.uint64)
        constants[ps.num_selectors, 0], constants[ps.num_selectors + 1, 0] = 1, 0            # row 0: x * y
        constants[ps.num_selectors, 1], constants[ps.num_selectors + 1, 1] = 1 << n_log, 1   # row 1: 2^20 * m0 * m1 + addend
        copies, presets, gens = [], {}, []
        cp = lambda a, b: copies.append((a[0] * n + a[1], b[0] * n + b[1]))
        # numerator and denominator live on a noop row (virtual targets of the gadget)
        for i in range(2):
            presets[(i, 5)], presets[(2 + i, 5)] = num[i], den[i]
            cp((2 + i, 5), (i, 0))                       # op 0 of row 0: x = den
            presets[(4 + i, 0)] = 0                      # addend
            cp((6 + i, 0), (i, 5))                       # ... and its output is the numerator
        gens.append(("quotient_ext", 0, [(0, 5), (1, 5), (2, 5), (3, 5)], [(2, 0), (3, 0)]))
        for op in range(1, ax.p0):                       # unused operations
            for k in range(6):
                presets[(8 * op + k, 0)] = 0
        # split_low_high(x, 20): low / high are fresh targets; row 1 op 0 recomposes them
        presets[(10, 6)] = x
        gens.append(("low_high", n_log, [(10, 6)], [(2, 1), (0, 1)]))     # low -> addend, high -> m0
        presets[(1, 1)] = 1
        cp((3, 1), (10, 6))
        gens.append(("copy", 0, [(0, 1)], [(20, 6)]))                     # a copy of `high` on the noop row
        for op in range(1, ar.p0):
            for k in range(3):
                presets[(4 * op + k, 1)] = 0
        circ = api.Circuit(ps, log_n, row_gate, constants, copies, generators=gens)
        constants[:ps.num_selectors] = circ.selector_columns()
        circ = api.Circuit(ps, log_n, row_gate, constants, copies, generators=gens)
        w = circ.generate_witness(presets)
        ok, msg = circ.check_witness(w, [0, 0, 0, 0])
        assert ok, msg
        q = (int(w[2, 0]), int(w[3, 0]))
        want = ((den[0] * q[0] + 7 * den[1] * q[1]) % P, (den[0] * q[1] + den[1] * q[0]) % P)
        assert want == tuple(num)
        assert int(w[2, 1]) == x & ((1 << n_log) - 1) and int(w[0, 1]) == x >> n_log and int(w[20, 6]) == x >> n_log
        plan = circ.witness_plan(list(presets))
        assert (plan.run(list(presets.values())) == w).all()
    # division by zero is an error of the run, not of the plan
    presets[(2, 5)] = presets[(3, 5)] = 0
    with pytest.raises(api.VpbsError, match="division by zero"):
        circ.generate_witness(presets)


def test_malformed_circuit_descriptions_are_errors():
    ps = api.GateSet(["noop", "arithmetic"])
    n = 8
    ok_rows = np.zeros(n, np.uint32)
    consts = np.zeros((ps.num_selectors + 2, n), np.uint64)
    for bad in (dict(row_gate=np.full(n, 7, np.uint32)),                      # gate index out of range
                dict(copies=[(80 * n, 0)]),                                    # copy constraint on a non-routed wire
                dict(generators=[("equality", 0, [(0, 0)], [(1, 0), (2, 0)])]),  # EqualityGenerator needs two inputs
                dict(generators=[("wire_split", 64, [(0, 0)], [(1, 0)])]),     # more than 63 bits per BaseSumGate
                dict(generators=[("quotient_ext", 0, [(0, 0), (1, 0)], [(2, 0), (3, 0)])]),   # needs numerator and denominator
                dict(generators=[("low_high", 64, [(0, 0)], [(1, 0), (2, 0)])]),               # n_log out of range
                dict(generators=[("base_sum", 2, [(200, 0)], [(1, 0)])])):     # position beyond the trace
        kw = dict(row_gate=ok_rows, copies=[], generators=())
        kw.update(bad)
        circ = api.Circuit(ps, 3, kw["row_gate"], consts, kw["copies"], generators=kw["generators"])
        with pytest.raises(api.VpbsError):
            circ.sigma_values()
        with pytest.raises(api.VpbsError):
            circ.generate_witness({})
        with pytest.raises(api.VpbsError):
            circ.check_witness(np.zeros((135, n), np.uint64), [0] * 4)
    # too few constants columns for the gate set
    circ = api.Circuit(ps, 3, np.full(n, ps.by_kind("arithmetic").index, np.uint32), np.zeros((1, n), np.uint64), [])
    with pytest.raises(api.VpbsError, match="constants columns"):
        circ.generate_witness({})
