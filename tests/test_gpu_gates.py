"""GPU parity tests (-m gpu) of the gate-constraint kernels: the HIP path through the C ABI against
oracle/gates.c on identical inputs -- bit-exact -- and complete proofs of a circuit with gates and copy constraints."""
import random

import numpy as np
import pytest

import gates_oracle as go
import oracle as orc
import step_oracle
import vpbs_amd
from vpbs_amd import api, synth

pytestmark = pytest.mark.gpu
P = api.P
DIGEST = [0x1111, 0x2222, 0x3333, 0x4444]
rng = np.random.default_rng(20241002)
ALL = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing",
       "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"]
SMALL_VARIANTS = [("base_sum", 10, 3), ("random_access", 1), ("random_access", 2), ("random_access", 3), ("random_access", 5),
                  ("coset_interpolation", 2), ("coset_interpolation", 3), ("coset_interpolation", 5), ("constant", 1), ("reducing", 5),
                  ("reducing_ext", 1), ("exponentiation", 7), ("mul_ext", 2), ("arithmetic", 3)]


def rand_field(*shape):
    return rng.integers(0, P, size=shape, dtype=np.uint64)


@pytest.fixture(scope="module")
def ctx():
    c = vpbs_amd.Context(0, log_n_max=16)
    yield c
    c.close()


def _leaf_to_natural(leaf_order, log_big):
    idx = np.array([int(format(t, "0%db" % log_big)[::-1], 2) for t in range(1 << log_big)])
    return np.ascontiguousarray(leaf_order[:, idx])   # natural[t] = leaf[bitrev(t)]


def _device_gate_terms(ctx, cs, wb, ps, pi_hash, alphas):
    import torch
    big = wb.n * 8
    out = torch.zeros((len(alphas), big), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.gate_terms(cs, wb, ps, pi_hash, alphas, out.data_ptr())
    ctx.synchronize()
    return out.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("log_n,spec,nc", [(3, ALL, 2), (6, ALL, 2), (5, ALL[:6], 1), (4, ["poseidon", "noop"], 3), (5, SMALL_VARIANTS[:7] + ["noop"], 2),
                                           (4, SMALL_VARIANTS[7:] + ["poseidon_mds"], 4)])
def test_gate_terms_match_oracle(ctx, log_n, spec, nc):
    """random wires, random constants AND random selector columns (every filter non-zero, so every gate of the set contributes
    at every point): device [nc][8n] (leaf order) == oracle (natural order), bit for bit"""
    n = 1 << log_n
    gs, ps = go.GateSet(spec), api.GateSet(spec)
    n_const = gs.num_selectors + gs.num_constants
    consts, wires = rand_field(n_const + 3, n), rand_field(135, n)   # + 3 columns standing in for sigmas
    wires[:, 1], wires[:, 2] = P - 1, 0                              # trace rows of boundary values (the LDE of the columns stays generic)
    wires[::2, 3], wires[1::2, 3] = (1 << 32) - 1, P - (1 << 32)
    pi_hash, alphas = [int(x) for x in rand_field(4)], [int(x) for x in rand_field(nc)]
    cs, wb = ctx.commit_values(consts), ctx.commit_values(wires)
    got = _leaf_to_natural(_device_gate_terms(ctx, cs, wb, ps, pi_hash, alphas), log_n + 3)
    want = gs.terms_coset(cs.coeffs()[:n_const], wb.coeffs(), pi_hash, alphas)
    assert got.shape == want.shape and (got == want).all()
    cs.free(); wb.free()


def test_gate_terms_vanish_on_a_satisfying_trace(ctx):
    """On a valid witness the device terms are divisible by Z_H: feeding them (alone: 1 routed wire with the identity sigma
    contributes a trivially satisfied permutation term) through the quotient kernel gives chunks whose evaluation at a random point
    matches terms(zeta) / Z_H(zeta) as computed by the oracle from the openings."""
    r = random.Random(5)
    log_n = 6
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    pis = [r.randrange(P) for _ in range(4)]
    constants, wires, sigma, pi_hash = go.demo_circuit(r, gs, log_n, pis)
    cs, wb = ctx.commit_values(np.concatenate([constants, sigma])), ctx.commit_values(wires)
    alphas = [r.randrange(P), r.randrange(P)]
    got = _leaf_to_natural(_device_gate_terms(ctx, cs, wb, ps, pi_hash, alphas), log_n + 3)
    want = gs.terms_coset(cs.coeffs()[:constants.shape[0]], wb.coeffs(), pi_hash, alphas)
    assert (got == want).all()
    cs.free(); wb.free()


@pytest.mark.parametrize("log_n", [6, 9])
def test_gate_circuit_proof_end_to_end(ctx, log_n):
    """A complete proof, every prover stage on the GPU, of a circuit with an in-circuit public-input hash, a Poseidon chain,
    chained arithmetic ops, rows of all 14 gate types and their copy constraints.  Bit-exact against the oracle prover (log_n = 6),
    accepted by the oracle verifier and by the product's host verifier with the gate constraints evaluated at zeta; a witness
    with one wrong value is rejected by both."""
    r = random.Random(100 + log_n)
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    n_routed = 80
    pis = [r.randrange(P) for _ in range(4)]
    constants, wires, sigma, pi_hash = go.demo_circuit(r, gs, log_n, pis)
    n_constants = constants.shape[0]
    ncols = [n_constants + n_routed, 135, 20, 16]
    cs_values = np.concatenate([constants, sigma])

    def run(w):
        cs = ctx.commit_values(cs_values)
        si = ctx.make_step_inputs(log_n, w, None, None, cs, DIGEST, pis, sigmas=sigma, n_routed=n_routed, n_constants=n_constants, gates=ps)
        proof = ctx.prove_step(si)
        cap = cs.cap()
        cs.free()
        return proof, cap

    def oracle_accepts(proof):
        op = proof["openings"]
        n_cs, n_w, n_z, n_q = ncols
        cs_z, w_z = op[:n_cs], op[n_cs:n_cs + n_w]
        zs_all, q_z, zs_next = op[n_cs + n_w:n_cs + n_w + n_z], op[n_cs + n_w + n_z:n_cs + n_w + n_z + n_q], op[n_cs + n_w + n_z + n_q:]
        ch = [int(x) for x in proof["challenges"]]
        betas, gammas, alphas, zeta = ch[0:2], ch[2:4], ch[4:6], ch[6:8]
        gt = gs.terms_zeta(cs_z[:n_constants], w_z, pi_hash, alphas)
        return orc.check_vanishing_at_zeta(w_z[:n_routed], cs_z[n_constants:], zs_all[:2], zs_next, zs_all[2:], q_z, log_n, betas, gammas,
                                           alphas, zeta, gate_terms_zeta=gt)

    proof, cap = run(wires)
    assert step_oracle.verify_step(proof, cap, ncols, DIGEST, pis, log_n)
    assert oracle_accepts(proof)
    assert api.verify_step(proof, cap, ncols, DIGEST, pis, log_n, check_permutation=True, n_constants=n_constants, n_routed=n_routed, gates=ps)
    assert not api.verify_step(proof, cap, ncols, DIGEST, pis, log_n, check_permutation=True, n_constants=n_constants, n_routed=n_routed)
    if log_n <= 6:
        want = step_oracle.prove_step({"constants_sigmas": cs_values, "wires": wires, "quotient": None}, DIGEST, pis, log_n, sigmas=sigma,
                                      n_routed=n_routed, n_constants=n_constants, gates=gs)
        for key in ("caps", "openings", "fri"):
            assert (proof[key] == want[key]).all(), key
    bad = wires.copy()
    bad[14, 3] = (int(bad[14, 3]) + 1) % P   # an output of a Poseidon row
    proof_bad, cap = run(bad)
    assert step_oracle.verify_step(proof_bad, cap, ncols, DIGEST, pis, log_n)   # commitments are consistent ...
    assert not oracle_accepts(proof_bad)                                          # ... but the quotient is not a polynomial identity
    assert not api.verify_step(proof_bad, cap, ncols, DIGEST, pis, log_n, check_permutation=True, n_constants=n_constants, n_routed=n_routed,
                               gates=ps)


@pytest.mark.parametrize("log_n", [15, 16])
def test_gate_terms_full_size_spot_check(ctx, log_n):
    """BASELINE config 2 shape (degree 2^15 / 2^16, LDE 2^18 / 2^19): all 14 gate types over random columns; the device value at
    sampled leaves equals the oracle's evaluation of the same gates on the LDE rows read back from the committed batches."""
    n = 1 << log_n
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    n_const = gs.num_selectors + gs.num_constants
    consts = synth.trace(0xC0DE, n_const, log_n)
    wires = synth.trace(0xC0DF, 135, log_n)
    pi_hash, alphas = [int(x) for x in rand_field(4)], [int(x) for x in rand_field(2)]
    cs, wb = ctx.commit_values(consts), ctx.commit_values(wires)
    got = _device_gate_terms(ctx, cs, wb, ps, pi_hash, alphas)   # leaf order
    for leaf in (0, 1, 77777, 8 * n - 1, 4 * n, 4 * n + 1, 8 * n - 77):
        c_row, w_row = cs.open(leaf)[0], wb.open(leaf)[0]   # MerkleTree::get: the LDE row at this leaf index
        cz = np.stack([c_row, np.zeros_like(c_row)], axis=1)
        wz = np.stack([w_row, np.zeros_like(w_row)], axis=1)
        want = gs.terms_zeta(cz, wz, pi_hash, alphas)
        assert (want[:, 1] == 0).all()
        assert [int(got[a][leaf]) for a in range(2)] == [int(want[a][0]) for a in range(2)]
    cs.free(); wb.free()


def test_gate_argument_errors_device(ctx):
    ps = api.GateSet(ALL)
    cs, wb = ctx.commit_values(rand_field(3, 16)), ctx.commit_values(rand_field(135, 16))
    import torch
    out = torch.zeros((2, 128), dtype=torch.int64, device="cuda")
    with pytest.raises(api.VpbsError):   # too few constants columns for the selectors + gate constants
        ctx.gate_terms(cs, wb, ps, [0] * 4, [1, 2], out.data_ptr())
    small = ctx.commit_values(rand_field(20, 16))
    cs2 = ctx.commit_values(rand_field(8, 16))
    with pytest.raises(api.VpbsError):   # PoseidonGate needs 135 wires
        ctx.gate_terms(cs2, small, ps, [0] * 4, [1, 2], out.data_ptr())


@pytest.mark.parametrize("log_n", [15, 16])
def test_full_size_gate_circuit_proof_verifies(ctx, log_n):
    """BASELINE config 2 size (degree 2^15 and 2^16, LDE 2^18 / 2^19) with a satisfiable circuit: public-input hash, Poseidon chain, arithmetic
    chain, rows of every gate type, copy constraints, 2^15 - 33 NoopGate rows.  The oracle prover would take minutes, so parity is
    carried by the identity itself: the GPU proof is accepted by the oracle's FRI verifier, the vanishing identity holds at zeta with
    the gate terms re-evaluated from the openings by the ORACLE (and by the product's host verifier); one wrong witness value
    anywhere breaks it."""
    r = random.Random(2024)
    n_routed = 80
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    pis = [r.randrange(P) for _ in range(4)]
    constants, wires, sigma, pi_hash = go.demo_circuit(r, gs, log_n, pis)
    n_constants = constants.shape[0]
    ncols = [n_constants + n_routed, 135, 20, 16]
    cs_values = np.concatenate([constants, sigma])

    def accepted(w):
        cs = ctx.commit_values(cs_values)
        si = ctx.make_step_inputs(log_n, w, None, None, cs, DIGEST, pis, sigmas=sigma, n_routed=n_routed, n_constants=n_constants, gates=ps)
        proof = ctx.prove_step(si)
        cap = cs.cap()
        cs.free()
        assert step_oracle.verify_step(proof, cap, ncols, DIGEST, pis, log_n)
        op = proof["openings"]
        n_cs, n_w, n_z, n_q = ncols
        cs_z, w_z = op[:n_cs], op[n_cs:n_cs + n_w]
        zs_all, q_z, zs_next = op[n_cs + n_w:n_cs + n_w + n_z], op[n_cs + n_w + n_z:n_cs + n_w + n_z + n_q], op[n_cs + n_w + n_z + n_q:]
        ch = [int(x) for x in proof["challenges"]]
        betas, gammas, alphas, zeta = ch[0:2], ch[2:4], ch[4:6], ch[6:8]
        gt = gs.terms_zeta(cs_z[:n_constants], w_z, pi_hash, alphas)
        ok = orc.check_vanishing_at_zeta(w_z[:n_routed], cs_z[n_constants:], zs_all[:2], zs_next, zs_all[2:], q_z, log_n, betas, gammas,
                                         alphas, zeta, gate_terms_zeta=gt)
        assert api.verify_step(proof, cap, ncols, DIGEST, pis, log_n, check_permutation=True, n_constants=n_constants, n_routed=n_routed,
                               gates=ps) == ok
        return ok

    assert accepted(wires)
    bad = wires.copy()
    bad[70, 2] = (int(bad[70, 2]) + 5) % P   # a partial-round S-box wire of a Poseidon row
    assert not accepted(bad)


def test_product_witness_to_proof(ctx):
    """Everything product-side: circuit data -> vpbs_selector_columns / vpbs_sigma_values / vpbs_generate_witness (host) -> step proof on
    the GPU -> vpbs_verify_step.  The Python circuit builder only supplies the circuit description and the PartialWitness."""
    import test_gates_cpu as tg
    r = random.Random(404)
    log_n, n_routed = 7, 80
    n = 1 << log_n
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    pis = [r.randrange(P) for _ in range(4)]
    constants, wires, _, pi_hash, desc = go.demo_circuit(r, gs, log_n, pis, describe=True)
    circ = api.Circuit(ps, log_n, desc["row_gate"], constants, desc["copies"])
    generated = set()
    for row in range(n):
        generated |= {(w, row) for w in tg._owned_wires(gs.gates[int(desc["row_gate"][row])])}
    fed = set()
    for cl in desc["classes"]:
        if any(tuple(x) in generated for x in cl):
            fed |= {tuple(x) for x in cl}
    presets = {}
    for row in range(n):
        g = gs.gates[int(desc["row_gate"][row])]
        if g.kind != "public_input":
            presets.update({(w, row): int(wires[w, row]) for w in tg._free_inputs(g) if (w, row) not in fed})
    witness = circ.generate_witness(presets)
    sigma = circ.sigma_values()
    cs_values = np.concatenate([circ.selector_columns(), constants[gs.num_selectors:], sigma])
    n_constants = constants.shape[0]
    ncols = [n_constants + n_routed, 135, 20, 16]
    cs = ctx.commit_values(cs_values)
    si = ctx.make_step_inputs(log_n, witness, None, None, cs, DIGEST, pis, sigmas=sigma, n_routed=n_routed, n_constants=n_constants, gates=ps)
    proof = ctx.prove_step(si)
    assert api.verify_step(proof, cs.cap(), ncols, DIGEST, pis, log_n, check_permutation=True, n_constants=n_constants, n_routed=n_routed, gates=ps)
    assert step_oracle.verify_step(proof, cs.cap(), ncols, DIGEST, pis, log_n)
    # other public inputs: the in-circuit hash no longer matches the PublicInputGate constraint
    assert not api.verify_step(proof, cs.cap(), ncols, DIGEST, [pis[0] ^ 1] + pis[1:], log_n, check_permutation=True, n_constants=n_constants,
                               n_routed=n_routed, gates=ps)
    cs.free()


def test_bsk_hash_subcircuit_of_the_step_circuit(ctx):
    """The bootstrapping-key hash of the reference's step circuit (/root/reference/src/vtfhe/ivc_based_vpbs.rs:126-133:
    current_bsk_hash_out = hash_n_to_hash_no_pad(current_bsk_hash_in || ggsw.flatten()), registered as public inputs) as a circuit of
    its own, at the paper's parameters (K = 2, ELL = 4, N = 1024: 4 + 16384 elements -> 2049 chained PoseidonGate rows, overwrite-mode
    sponge through copy constraints).  Witness generated by the product (vpbs_generate_witness), proof on the GPU, verified by the
    product's verifier; the public inputs equal the native chain hash of verify_hash_output (ivc_based_vpbs.rs:64-78)."""
    spec = ["noop", "public_input", "poseidon"]
    gs, ps = go.GateSet(spec), api.GateSet(spec)
    K, ELL, N = 2, 4, 1024
    item = synth.field_elements(0xB5C, K * ELL * K * N)             # Ggsw::flatten() of one bootstrapping-key element
    h_in = np.zeros(4, np.uint64)                                    # current_bsk_hash_in of the first CMUX step
    data = np.concatenate([h_in, item])
    n_chunks = (data.size + 7) // 8
    log_n = 12
    n = 1 << log_n
    pos_gate, pi_gate, noop = ps.by_kind("poseidon"), ps.by_kind("public_input"), ps.by_kind("noop")
    row_gate = np.full(n, noop.index, np.uint32)
    row_gate[0] = pi_gate.index
    row_gate[1:2 + n_chunks] = pos_gate.index                       # rows 1..n_chunks: the sponge; row n_chunks + 1: public-input hash
    presets, copies = {}, []
    P_ = lambda c, r: c * n + r
    for k in range(n_chunks):
        r = 1 + k
        chunk = data[8 * k:8 * k + 8]
        for i in range(12):
            if i < chunk.size:
                presets[(i, r)] = int(chunk[i])                      # overwrite mode: the new block
            elif k == 0:
                presets[(i, r)] = 0                                  # initial state
            else:
                copies.append((P_(12 + i, r - 1), P_(i, r)))        # the rest of the state carries over
        presets[(24, r)] = 0                                         # swap
    r_pi = 1 + n_chunks
    for i in range(12):
        if i < 4:
            copies.append((P_(12 + i, r_pi - 1), P_(i, r_pi)))      # public inputs = the hash output ...
            copies.append((P_(12 + i, r_pi), P_(i, 0)))             # ... and their hash feeds the PublicInputGate
        else:
            presets[(i, r_pi)] = 0
    presets[(24, r_pi)] = 0
    constants = np.zeros((ps.num_selectors + max(1, ps.num_constants), n), np.uint64)
    circ = api.Circuit(ps, log_n, row_gate, constants, copies)
    constants[:ps.num_selectors] = circ.selector_columns()
    circ = api.Circuit(ps, log_n, row_gate, constants, copies)
    wires = circ.generate_witness(presets)
    out = wires[12:16, r_pi - 1]
    native, ok = api.hash_chain(item[None, :], claimed=out)         # verify_hash_output with one item
    assert ok and (native == out).all() and (native == orc.hash_no_pad(data)).all()
    pis = [int(x) for x in out]
    assert [int(wires[i, 0]) for i in range(4)] == [int(x) for x in api.hash_no_pad(out)]
    sigma = circ.sigma_values()
    n_constants, n_routed = constants.shape[0], 80
    cs = ctx.commit_values(np.concatenate([constants, sigma]))
    si = ctx.make_step_inputs(log_n, wires, None, None, cs, DIGEST, pis, sigmas=sigma, n_routed=n_routed, n_constants=n_constants, gates=ps)
    proof = ctx.prove_step(si)
    ncols = [n_constants + n_routed, 135, 20, 16]
    assert api.verify_step(proof, cs.cap(), ncols, DIGEST, pis, log_n, check_permutation=True, n_constants=n_constants, n_routed=n_routed, gates=ps)
    assert step_oracle.verify_step(proof, cs.cap(), ncols, DIGEST, pis, log_n)
    wrong = list(pis); wrong[2] ^= 1
    assert not api.verify_step(proof, cs.cap(), ncols, DIGEST, wrong, log_n, check_permutation=True, n_constants=n_constants,
                               n_routed=n_routed, gates=ps)
    cs.free()


def test_cxx_circuit_example(ctx):
    """examples/prove_bsk_hash.cpp: the same bootstrapping-key hash circuit driven from plain C++ through the C ABI only (layout,
    selector columns, sigma values, witness generation, commit, prove with gates, verify): exits 0 and prints the hash that
    vpbs_hash_chain / the oracle compute natively."""
    import subprocess
    import __graft_entry__ as entry
    exe = entry.build_example("prove_bsk_hash")
    r = subprocess.run([exe, "2", "2", "64"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "proof verified: 1; with a wrong public input: 0" in r.stdout
    item = synth.field_elements(0xB5C, 2 * 2 * 2 * 64)
    want = orc.hash_no_pad(np.concatenate([np.zeros(4, np.uint64), item]))
    assert " ".join("%016x" % int(x) for x in want) in r.stdout


def test_product_reproduces_frozen_step_proofs(ctx):
    """the same regression vectors through the HIP path (C ABI): bit-identical caps, challenges, openings and FRI proofs -- including the
    instance bench.py times (degree 2^16, 135/20/16/86 columns, all 14 gate types, 4173 public inputs: caps, challenges, openings, FRI
    words and the serialised bytes frozen from the oracle's proof)"""
    import regression_cases as rc
    for case in rc.cases(full_size=True):
        b = rc.build(case)
        digest = b.get("digest", rc.DIGEST)
        cs = ctx.commit_values(b["inputs"]["constants_sigmas"])
        if b["gates"] is None:
            si = ctx.make_step_inputs(b["log_n"], b["inputs"]["wires"], b["inputs"]["zs_partial_products"], b["inputs"]["quotient"], cs, digest,
                                      b["pis"])
        else:
            si = ctx.make_step_inputs(b["log_n"], b["inputs"]["wires"], None, None, cs, digest, b["pis"], sigmas=b["sigma"], n_routed=80,
                                      n_constants=b["n_constants"], gates=api.GateSet(b["gates"]))
        proof = ctx.prove_step(si)
        rc.check(case, proof)
        if "cs_cap_sha256" in case:
            assert rc.sha(cs.cap()) == case["cs_cap_sha256"]
        rc.check_bytes(case, ctx.step_proof_to_bytes(si, b["n_constants"], proof))
        cs.free()


def test_full_size_step_proof_bit_exact_against_the_oracle(ctx):
    """BASELINE config 2 at the benchmarked shape -- degree 2^16, LDE 2^19, 135 wire / 20 Z+pp / 16 quotient / 86 constant+sigma
    columns, the constraints of all 14 gate types, partial products and quotient computed by the prover, 4173 public inputs -- proven by
    the HIP path and by the C oracle (run here, ~10-40 s of the host's cores) on the same seeded inputs: every word must agree, and the
    serialised bytes (ivc_based_vpbs.rs:488 to_bytes) too.  A second instance (other seeds) so that this is not the frozen one."""
    import regression_cases as rc
    import step_oracle
    B = rc.BENCH
    log_n, nc, nr = B["log_n"], B["n_constants"], B["n_routed"]
    inputs = synth.step_inputs(log_n, instance=3, cols=B["cols"])
    inputs["quotient"] = None
    pis = synth.field_elements(0xABCD + 3, B["n_public_inputs"])
    sig = np.ascontiguousarray(inputs["constants_sigmas"][nc:nc + nr])
    cs = ctx.commit_values(inputs["constants_sigmas"])
    si = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs, B["digest"], pis, sigmas=sig, n_routed=nr, n_constants=nc,
                              gates=api.GateSet(rc.GATES))
    got = ctx.prove_step(si)
    want = step_oracle.prove_step(inputs, B["digest"], pis, log_n, sigmas=sig, n_routed=nr, n_constants=nc, gates=go.GateSet(rc.GATES))
    assert (cs.cap() == want["cs_cap"]).all()
    for key in ("caps", "challenges", "openings", "fri"):
        assert (np.asarray(got[key]).reshape(-1) == np.asarray(want[key]).reshape(-1)).all(), key
    assert ctx.step_proof_to_bytes(si, nc, got) == step_oracle.to_bytes(want, want["ncols"], nc, pis, log_n)
    cs.free()


def test_gate_lanes_setting_does_not_change_the_proof(ctx):
    """one stream or three (gate kernels + permutation part overlapped): identical proof"""
    r = random.Random(11)
    log_n, n_routed = 8, 80
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    pis = [r.randrange(P) for _ in range(4)]
    constants, wires, sigma, _ = go.demo_circuit(r, gs, log_n, pis)
    cs = ctx.commit_values(np.concatenate([constants, sigma]))
    si = ctx.make_step_inputs(log_n, wires, None, None, cs, DIGEST, pis, sigmas=sigma, n_routed=n_routed, n_constants=constants.shape[0], gates=ps)
    proofs = []
    for lanes in (3, 1, 3):
        ctx.set_gate_lanes(lanes)
        proofs.append(ctx.prove_step(si))
    ctx.set_gate_lanes(1)   # the default
    for key in ("caps", "openings", "fri"):
        assert (proofs[0][key] == proofs[1][key]).all() and (proofs[0][key] == proofs[2][key]).all()
    with pytest.raises(api.VpbsError):
        ctx.set_gate_lanes(2)
    cs.free()


def test_device_witness_for_every_gate_type(ctx):
    """vpbs_witness_device_* on the demo circuit with rows of all 14 gate types (in-circuit public-input hash, Poseidon and arithmetic
    chains through copy constraints, interpolation / random-access / exponentiation / reducing rows): two instances with different
    PartialWitnesses in one batch, each identical to the host plan's witness and satisfying every constraint."""
    import torch
    import test_gates_cpu as tg
    log_n = 7
    n = 1 << log_n
    gs, ps = go.GateSet(ALL), api.GateSet(ALL)
    columns, witnesses, circ, positions = [], [], None, None
    for seed in (404, 405):
        r = random.Random(404)                      # the same circuit ...
        pis = [random.Random(seed).randrange(P) for _ in range(4)]   # ... proving other public inputs
        constants, wires, _, pi_hash, desc = go.demo_circuit(r, gs, log_n, pis, describe=True)
        if circ is None:
            circ = api.Circuit(ps, log_n, desc["row_gate"], constants, desc["copies"])
            base_constants = constants
        assert (constants == base_constants).all() and (desc["row_gate"] == circ.row_gate).all()   # same circuit, other values
        generated = set()
        for row in range(n):
            generated |= {(w, row) for w in tg._owned_wires(gs.gates[int(desc["row_gate"][row])])}
        fed = set()
        for cl in desc["classes"]:
            if any(tuple(x) in generated for x in cl):
                fed |= {tuple(x) for x in cl}
        presets = {}
        for row in range(n):
            g = gs.gates[int(desc["row_gate"][row])]
            if g.kind != "public_input":
                presets.update({(w, row): int(wires[w, row]) for w in tg._free_inputs(g) if (w, row) not in fed})
        if positions is None:
            positions = list(presets)
        assert list(presets) == positions
        columns.append([presets[p] for p in positions])
        witnesses.append((wires, pi_hash))
    plan = circ.witness_plan(positions)
    st = plan.stats()
    assert st["levels"] > 0
    dev = api.WitnessDevice(ctx, plan, max_batch=2)
    dev.run(np.ascontiguousarray(np.array(columns, dtype=np.uint64).T))
    d_w = torch.zeros((135, n), dtype=torch.int64, device="cuda")
    for i in range(2):
        dev.wires(i, d_w.data_ptr())
        got = d_w.cpu().numpy().view(np.uint64)
        want = plan.run(columns[i])
        assert (got == want).all(), np.argwhere(got != want)[:5]
        ok, msg = circ.check_witness(got, witnesses[i][1])
        assert ok, msg
    assert not (np.array(columns[0]) == np.array(columns[1])).all()
    dev.free(); plan.free()
