"""The reference's CYCLIC step circuit (ivc_based_vpbs.rs:159-386: each step proof verifies the previous one in circuit) on the CPU: the
circuit description of circuitgen/cyclic_circuit.py, witnesses by the PRODUCT's generators (host), proofs by the CPU oracle's prover, every
proof accepted by the product's host verifier.  The GPU twin is tests/test_gpu_step_circuit.py::test_ivc_chain_*."""
import os
import random

import numpy as np
import pytest

import cyclic_circuit as cc
import gates_oracle as go
import oracle as orc
import step_oracle
import tfhe_oracle as T
from cyclic_circuit import P
from vpbs_amd import api


def test_gate_constraints_model_matches_the_product():
    """the gate-constraint restatement the in-circuit verifier is generated from (cyclic_circuit.gate_constraints, all 14 gates) equals the
    product's evaluation at a GF(p^2) point (vpbs_gate_terms_at)"""
    rnd = random.Random(9)
    gs = api.GateSet(cc.GATE_SPEC)
    consts = [(rnd.randrange(P), rnd.randrange(P)) for _ in range(gs.num_selectors + gs.num_constants)]
    wires = [(rnd.randrange(P), rnd.randrange(P)) for _ in range(135)]
    pih = [rnd.randrange(P) for _ in range(4)]
    alphas = [rnd.randrange(P) for _ in range(2)]
    B = cc.NumBackend()
    total = cc.gate_terms(B, gs, api, consts, wires, [(h, 0) for h in pih])
    want = gs.terms_at(np.array(consts, np.uint64), np.array(wires, np.uint64), pih, alphas)
    for a in range(2):
        acc = (0, 0)
        for t in reversed(total):
            acc = B.add(B.mul(acc, (alphas[a], 0)), t)
        assert acc == tuple(int(x) for x in want[a])


class OracleProver:
    """prove / verify of one circuit with the CPU oracle (the GPU tests use the product's prover here)"""

    def __init__(self, built):
        self.built = built
        self.sigma = built.circuit.sigma_values()
        self.cs_values = np.concatenate([built.constants, self.sigma])
        self.cs = orc.Batch(self.cs_values, 3, 4, True)
        self.cap = self.cs.cap()
        self.vk = cc.vk_words(self.cap, built.log_n)
        self.nconst = built.constants.shape[0]
        self.gs, self.ps = go.GateSet(cc.GATE_SPEC), api.GateSet(cc.GATE_SPEC)

    def prove(self, wires, pis):
        p = step_oracle.prove_step({"constants_sigmas": self.cs_values, "wires": wires, "quotient": None}, self.vk[:4], pis, self.built.log_n,
                                   cs_batch=self.cs, sigmas=self.sigma, n_routed=80, n_constants=self.nconst, gates=self.gs)
        assert self.verify(p, pis)
        return p

    def verify(self, p, pis):
        return api.verify_step(p, self.cap, p["ncols"], self.vk[:4], pis, self.built.log_n, n_constants=self.nconst, n_routed=80, gates=self.ps)


def test_in_circuit_verifier_accepts_a_proof_and_rejects_a_tampered_one():
    """verify_proof (recursive_verifier.rs) as a circuit: the witness generated from a real proof satisfies every gate and copy constraint,
    the in-circuit transcript reproduces the prover's challenges; one flipped opening makes the circuit unsatisfiable"""
    rnd = random.Random(3)
    log_c = 6
    gs = go.GateSet(cc.GATE_SPEC)
    cpis = [rnd.randrange(P) for _ in range(4)]
    constants, wires, sigma, _ = go.demo_circuit(rnd, gs, log_c, cpis)
    cs_values = np.concatenate([constants, sigma])
    cs = orc.Batch(cs_values, 3, 4, True)
    digest = cc.circuit_digest(cs.cap(), log_c)
    proof = step_oracle.prove_step({"constants_sigmas": cs_values, "wires": wires, "quotient": None}, digest, cpis, log_c, cs_batch=cs, sigmas=sigma,
                                   n_routed=80, n_constants=constants.shape[0], gates=gs)
    shape = cc.Shape(api, log_c, 4)
    assert shape.ncols == proof["ncols"] and shape.fri_words == proof["fri"].size
    vc = cc.VerifierOnlyCircuit(api, shape)
    w = vc.built.circuit.generate_witness(vc.presets(shape.flat_proof(proof), cpis, digest, proof["cs_cap"]))
    ok, msg = vc.built.circuit.check_witness(w, api.hash_no_pad(np.array(vc.built.values(w, vc.built.public_inputs), np.uint64)))
    assert ok, msg
    ch = vc.challenges
    assert vc.built.values(w, ch["betas"] + ch["gammas"] + ch["alphas"] + list(ch["zeta"])) == [int(x) for x in proof["challenges"]]
    # EVERY word of the proof is bound: a few chosen positions, then a random sweep over caps, openings, query leaves and paths, fold
    # evaluations, final polynomial and proof-of-work witness (a word the in-circuit verifier did not constrain would pass here)
    plan = vc.built.circuit.witness_plan(list(vc.presets(shape.flat_proof(proof), cpis, digest, proof["cs_cap"])))
    vals = lambda flat: np.array(list(vc.presets(flat, cpis, digest, proof["cs_cap"]).values()), np.uint64)
    assert (plan.run(vals(shape.flat_proof(proof))) == w).all()
    sweep = [shape.caps_words + 10, 3, shape.proof_words - 1, shape.caps_words + shape.openings_words + 200] + \
            [rnd.randrange(shape.proof_words) for _ in range(400)]
    for at in sweep:
        bad = shape.flat_proof(proof).copy()
        bad[at] ^= np.uint64(1 << rnd.randrange(0, 40)) if at != sweep[0] else np.uint64(1)
        with pytest.raises(api.VpbsError, match="set twice|too large"):     # a connect that cannot hold, or the proof-of-work range check
            plan.run(vals(bad))
    plan.free()


def run_chain(cy, dm, C, D, prove_c, prove_d, keys, ct, acc_init, check=True):
    """verified_pbs (ivc_based_vpbs.rs:277-371): base proof of the dummy circuit, then the n + 2 steps, each taking the previous proof"""
    N, K, ELL, LOGB, n_lwe = cy.params
    s_to, s_lwe, s_glwe, bsk, ksk = keys
    flat = lambda acc: [int(v) for p in acc for v in p]
    base_pis = np.array(flat(acc_init) + [0] + [0] * (K * N) + [0] * 8 + [int(v) for v in C.vk], np.uint64)
    proof, pis = prove_d(dm.witness(base_pis), base_pis), base_pis
    # the second slot (dummy_proof_and_vk): the dummy circuit's proof of all-zero public inputs, the same in every step
    zero_pis = np.zeros(base_pis.size, np.uint64)
    dummy_flat = cy.shape.flat_proof(prove_d(dm.witness(zero_pis), zero_pis))
    plan = cy.built.circuit.witness_plan(cy.positions)
    zero_ggsw = np.zeros(K * ELL * K * N, np.uint64)
    steps = [(0, zero_ggsw, ct[n_lwe])] + [(1, bsk[x], ct[x]) for x in range(n_lwe)] + [(1, ksk, 0)]
    proofs = []
    # two-phase plan: the proof words are the late part of the PartialWitness (vpbs_witness_plan_split); both forms give the same wires
    split = cy.built.circuit.witness_plan(cy.positions)
    late = np.zeros(len(cy.positions), np.uint8)
    late[:cy.shape.proof_words] = 1
    split.split(late)
    # the late phase in STAGES (the sections of the previous proof in the order the prover finishes them: caps + openings | FRI commit caps,
    # final polynomial, proof-of-work witness | query rounds): stages run ahead one by one give the wires of the one-stage plan
    staged = cy.built.circuit.witness_plan(cy.positions)
    sh = cy.shape
    R = len(sh.arity_bits)                      # the sections vpbs_prove_step reports (vpbs_step_inputs.on_section): R + 3 stages
    stage_of = np.zeros(len(cy.positions), np.uint8)
    stage_of[:sh.proof_words] = R + 3
    stage_of[:sh.caps_words + sh.openings_words] = 1
    fri0 = sh.caps_words + sh.openings_words
    for r in range(R):
        stage_of[fri0 + r * 4 * sh.cap_len:fri0 + (r + 1) * 4 * sh.cap_len] = 2 + r
    stage_of[fri0 + sh.fri_words - 1 - 2 * sh.final_len:fri0 + sh.fri_words] = 2 + R
    staged.split(stage_of)
    assert staged.late_stages() == R + 3 and split.late_stages() == 1
    assert sorted(staged.late_positions()) == sorted(split.late_positions())       # the same wires, ordered by stage in the staged plan
    assert (staged.late_input_positions() == split.late_input_positions()).all()
    for cond, ggsw, mask in steps:
        values = cy.values(cy.shape.flat_proof(proof), pis, cond, ggsw, mask, C.vk, D.vk, dummy_flat)
        wires = plan.run(values)
        for ahead in (0, 1, 2, R + 2, R + 3):   # how many stages run before run_late; packed in place on the way (odd counts) or at the end
            three = np.empty_like(wires)
            st3 = staged.run_early(values, three)
            packed = np.full(staged.late_positions().size, 0xDEAD, np.uint64) if ahead % 2 else None
            for k in range(1, ahead + 1):
                partial = values.copy()
                partial[:sh.proof_words][stage_of[:sh.proof_words] > k] = 0xBAD   # words of later stages do not exist yet
                staged.run_late_stage(st3, k, partial, packed)
            assert (staged.run_late_packed(st3, values, packed) == wires.reshape(-1)[staged.late_positions()]).all(), ahead
        if cond and len(proofs) == 1:
            # Two states of ONE plan with stages run ahead: the stage before the last leaves the late pool's workers spinning for
            # "its" state.  B takes the pool after A was promised it; freeing A then must not put the workers of B's runs to sleep (A holds a
            # token the later run invalidated, not the pool), and B's late phase completes with the right wires.
            ma, mb = np.empty_like(wires), np.empty_like(wires)
            st_a = staged.run_early(values, ma)
            for k in range(1, R + 3):
                staged.run_late_stage(st_a, k, values)
            st_b = staged.run_early(values, mb)
            for k in range(1, R + 3):
                staged.run_late_stage(st_b, k, values)
            api.lib().vpbs_witness_state_free(st_a)
            assert (staged.run_late_packed(st_b, values) == wires.reshape(-1)[staged.late_positions()]).all()
            # ... and a state that outlives a re-split of its plan still frees cleanly (it shares the pool's ownership)
            other = cy.built.circuit.witness_plan(cy.positions)
            other.split(stage_of)
            st_c = other.run_early(values, ma)
            for k in range(1, R + 3):
                other.run_late_stage(st_c, k, values)
            other.split(late)                      # one-stage split: the staged pools are dropped
            api.lib().vpbs_witness_state_free(st_c)
            other.free()
        if cond and len(proofs) < 3:
            # A wrong word of a section is noticed by the stage that reads it (the first section by the first stage already: the transcript and
            # the cap connections are there) -- and the failure is the STATE's from then on: whether a slot mismatch or a generator
            # that rejected its inputs (a gadget's range check, a PoseidonGate swap wire that is no bit), every later stage reports it again and
            # run_late_packed fails instead of returning a witness built on a half-run stage.
            tamper = {"wires cap": 5, "an opening": sh.caps_words + 3, "first FRI cap": fri0 + 1,
                      "final polynomial": fri0 + sh.fri_words - 1 - 2 * sh.final_len, "proof-of-work witness": fri0 + sh.fri_words - 1}
            kinds = set()
            for what, at in tamper.items():
                bad = values.copy()
                bad[at] ^= np.uint64(1)
                st3 = staged.run_early(values, np.empty_like(wires))
                failed_at, first = None, ""
                for k in range(1, R + 3):
                    try:
                        staged.run_late_stage(st3, k, bad)
                    except api.VpbsError as e:
                        failed_at, first = k, str(e)
                        break
                if what in ("wires cap", "an opening"):
                    assert failed_at == 1, (what, failed_at)
                if failed_at is None:            # a word that only moves a challenge: the query rounds (the last stage, in run_late) notice
                    with pytest.raises(api.VpbsError):
                        staged.run_late_packed(st3, bad)
                    continue
                kinds.add(first.split(": ", 1)[-1][:40])
                if failed_at < R + 2:
                    with pytest.raises(api.VpbsError, match="had failed"):
                        staged.run_late_stage(st3, failed_at + 1, bad)
                with pytest.raises(api.VpbsError, match="had failed"):
                    staged.run_late_packed(st3, bad)                       # consumes the state
            assert kinds, kinds
            if os.environ.get("VPBS_TEST_VERBOSE"):
                print("late-stage failures seen:", kinds)
        early_values = values.copy()
        early_values[:cy.shape.proof_words] = 0xDEAD                  # the late entries are not read by the early phase
        two = np.empty_like(wires)
        state = split.run_early(early_values, two)
        seeded = split.state_from_late_inputs(two.reshape(-1)[split.late_input_positions()])   # what a device-side early phase hands over
        assert (split.run_late_packed(seeded, values) == wires.reshape(-1)[split.late_positions()]).all()
        assert (split.run_late(state, values, two) == wires).all()
        if cond:   # a proof with one flipped word makes the late phase fail (its generators run on several threads), whatever the word
            bad = values.copy()
            bad[(7919 * len(proofs) + 13) % cy.shape.proof_words] ^= np.uint64(1)
            state = split.run_early(early_values, two)
            with pytest.raises(api.VpbsError):
                split.run_late(state, bad, two)
        # the two proof slots (select_proof_with_pis): the verifier sees the slot `condition` selects and nothing of the other one
        W = cy.shape.proof_words
        at = (104729 * len(proofs) + 7) % W
        other_slot = values.copy()
        if cond:
            other_slot[len(values) - len(pis) - W + at] ^= np.uint64(1)          # a word of the dummy proof: not looked at
            in_slot = None
        else:
            other_slot[at] ^= np.uint64(1)                                       # base step: the cyclic slot's proof words are free ...
            in_slot = values.copy()
            in_slot[len(values) - len(pis) - W + at] ^= np.uint64(1)             # ... and the dummy proof is the one verified
        got = plan.run(other_slot)
        assert (cy.public_inputs(got) == cy.public_inputs(wires))
        if check and len(proofs) < 2:
            ok, msg = cy.built.circuit.check_witness(got, api.hash_no_pad(np.array(cy.public_inputs(got), np.uint64)))
            assert ok, msg
        if in_slot is not None:
            with pytest.raises(api.VpbsError):
                plan.run(in_slot)
        pis = np.array(cy.public_inputs(wires), np.uint64)
        if check:
            ok, msg = cy.built.circuit.check_witness(wires, api.hash_no_pad(pis))
            assert ok, msg
        proof = prove_c(wires, pis)
        proofs.append((proof, pis))
    plan.free()
    split.free()
    staged.free()
    return proofs


GOLDEN_CHAIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ivc_chain_n8.json")


def n8_chain_inputs():
    """keys, ciphertext and test vector of the N = 8, n = 1 chain both the CPU test (oracle prover) and the GPU test (vpbs_ivc_prove_pbs) run"""
    N, K, ELL, LOGB, n_lwe = 8, 2, 4, 5, 1
    rng = np.random.default_rng(5)
    ring = T.Ring(3)
    s_to, s_lwe, s_glwe, bsk, ksk = T.pbs_setup(ring, rng, n_lwe, K, ELL, LOGB)
    delta = T.get_delta(4)
    testv = T.get_testv(ring, 2, delta)
    ct = T.lwe_encrypt(rng, s_lwe, delta % P)
    return ring, (s_to, s_lwe, s_glwe, bsk, ksk), delta, testv, ct


def test_ivc_chain_on_the_cpu():
    """BASELINE config 1 (N = 8 ring, one blind-rotation step, CPU prover): the cyclic circuit fits degree 2^13; base proof + first step +
    CMUX + key switch, every proof verifying its predecessor in circuit; the last proof alone carries the statement (verify_pbs, :388-489):
    test vector, counter n + 2, accumulator = the native chain's, chain hashes = the native sponge, verifier data = the circuit's own."""
    N, K, ELL, LOGB, n_lwe, log_n = 8, 2, 4, 5, 1, 13
    cy = cc.CyclicStepCircuit(api, N, K, ELL, LOGB, n_lwe, orc.negacyclic_params(3), log_n)
    assert cy.built.log_n == 13 and cy.shape.n_pi == 109
    with pytest.raises(ValueError, match="does not fit"):
        cc.CyclicStepCircuit(api, N, K, ELL, LOGB, n_lwe, orc.negacyclic_params(3), 12)
    dm = cc.DummyCircuit(api, log_n, cy.shape.n_pi)
    C, D = OracleProver(cy.built), OracleProver(dm.built)
    ring, (s_to, s_lwe, s_glwe, bsk, ksk), delta, testv, ct = n8_chain_inputs()
    acc_init = [[0] * N for _ in range(K - 1)] + [testv]
    keys = (s_to, s_lwe, s_glwe, [T.flatten_ggsw(g) for g in bsk], T.flatten_ggsw(ksk))
    proofs = run_chain(cy, dm, C, D, C.prove, D.prove, keys, ct, acc_init)
    accs = T.pbs_chain(ring, acc_init, ct, bsk, ksk, K, ELL, LOGB)
    kn = K * N
    for s, (proof, pis) in enumerate(proofs):
        assert int(pis[kn]) == s + 1 and [int(v) for v in pis[kn + 1:2 * kn + 1]] == [int(v) for p in accs[s] for v in p]
    proof, pis = proofs[-1]
    # verify_pbs on the LAST proof only
    assert C.verify(proof, pis)
    assert [int(v) for v in pis[:kn]] == [0] * (kn - N) + [int(v) for v in testv] and int(pis[kn]) == n_lwe + 2
    assert (pis[-68:] == C.vk).all()                                                      # check_cyclic_proof_verifier_data
    zero_ggsw = np.zeros(K * ELL * K * N, np.uint64)
    bsk_items = np.stack([zero_ggsw] + keys[3] + [keys[4]])
    lwe_items = np.array([[ct[n_lwe]]] + [[ct[x]] for x in range(n_lwe)] + [[0]], np.uint64)
    assert api.hash_chain(bsk_items, pis[2 * kn + 1:2 * kn + 5])[1] and api.hash_chain(lwe_items, pis[2 * kn + 5:2 * kn + 9])[1]
    m_bar = T.glwe_decrypt(ring, s_to, [[int(v) for v in pis[kn + 1 + p * N:kn + 1 + (p + 1) * N]] for p in range(K)], K)
    assert round(m_bar[0] / delta) % 4 == 1
    # a proof of another statement is not accepted in its place
    wrong = pis.copy()
    wrong[kn + 2] ^= np.uint64(1)
    assert not C.verify(proof, wrong)
    # the same statement through the product's one-call verify_pbs (vpbs_verify_pbs) on the serialised proof, and each of its checks failing
    blob = step_oracle.to_bytes(proof, proof["ncols"], C.nconst, pis, log_n)
    out_ct = pis[kn + 1:2 * kn + 1]
    bsk_flat = np.stack(keys[3])

    def vp(blob=blob, testv=testv, ct=ct, bsk=bsk_flat, ksk=keys[4], out_ct=out_ct, cap=C.cap, digest=C.vk[:4]):
        return api.verify_pbs(blob, cap, proof["ncols"], digest, log_n, C.nconst, 80, C.ps, N, K, testv, ct, bsk, ksk, out_ct=out_ct)

    assert vp() == (True, "")
    with pytest.raises(api.VpbsError):       # the output ciphertext is part of the statement: no verdict without it
        vp(out_ct=None)
    # the chain is deterministic (smallest proof-of-work nonce): its last proof is frozen, and the GPU chain of the same inputs
    # (test_gpu_step_circuit.py::test_ivc_chain_bit_identical_to_the_cpu_oracle_chain) must produce the same bytes
    import hashlib
    import json
    digest = hashlib.sha256(blob).hexdigest()
    if os.environ.get("VPBS_RECORD_GOLDEN"):
        json.dump({"what": "sha256 of ProofWithPublicInputs::to_bytes of the LAST proof of the N = 8, n = 1 IVC chain (3 step proofs of the cyclic "
                           "circuit after the base proof; inputs: tests/test_cyclic_cpu.py::n8_chain_inputs), proven by the CPU oracle",
                   "bytes": len(blob), "sha256": digest}, open(GOLDEN_CHAIN, "w"), indent=1)
    frozen = json.load(open(GOLDEN_CHAIN))
    assert (len(blob), digest) == (frozen["bytes"], frozen["sha256"])
    other = lambda a, i=0: np.concatenate([np.asarray(a, np.uint64).reshape(-1)[:i], [np.uint64(int(np.asarray(a, np.uint64).reshape(-1)[i]) ^ 1)],
                                           np.asarray(a, np.uint64).reshape(-1)[i + 1:]]).astype(np.uint64)
    assert vp(testv=other(testv, 3)) == (False, "claimed test vector differs from testv")
    assert vp(out_ct=other(out_ct, 5)) == (False, "the output ciphertext is not the proof's accumulator")
    assert vp(ksk=other(keys[4], 7)) == (False, "the key hash chain does not match")
    assert vp(bsk=other(bsk_flat, 9).reshape(bsk_flat.shape)) == (False, "the key hash chain does not match")
    assert vp(ct=other(ct, 1)) == (False, "the LWE hash chain does not match")
    bad = bytearray(blob)
    bad[8 * 200] ^= 1                                                  # a word of the wires cap / openings
    assert vp(blob=bytes(bad))[0] is False
    assert vp(blob=blob[:-8])[1].startswith("the bytes are not a proof")
    assert vp(cap=D.cap, digest=D.vk[:4])[0] is False                  # the dummy circuit's verifier data: the proof does not verify there
    # an earlier proof of the chain: valid, but its counter is not n + 2
    p1, pis1 = proofs[1]
    assert vp(blob=step_oracle.to_bytes(p1, p1["ncols"], C.nconst, pis1, log_n), out_ct=pis1[kn + 1:2 * kn + 1]) == (False, "the counter is not n + 2")
