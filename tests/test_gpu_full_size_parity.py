"""Full-size parity of the REAL circuits against the CPU oracle: the reference-shaped circuits at the paper's
parameters -- the cyclic step circuit of verified_pbs (/root/reference/src/vtfhe/ivc_based_vpbs.rs:159-386; prove() at :333) with its real gates,
selectors, sigma polynomials and a witness that verifies the previous proof in circuit, and build_step_circuit (:80-157) -- proven once by the
HIP path through the C ABI and once by the C oracle, compared word for word: caps, challenges, openings, FRI words, serialised bytes.
(The other full-size comparisons in the suite run a synthetic 14-gate circuit; the chain at N = 8 is compared in test_gpu_step_circuit.py.)"""
import os
import sys

import numpy as np
import pytest

import gates_oracle as go
import oracle as orc
import step_oracle
import export_circuits
import vpbs_amd
from vpbs_amd import api, circuit_file

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, ELL, LOGB, N, N_LWE, LOG_N = 2, 4, 5, 1024, 728, 16
P = api.P


def oracle_gate_set(d):
    """the oracle's gate set from the gates of a circuit file (kind + parameters as exported)"""
    return go.GateSet([(api.GATE_KINDS[g.kind], g.p0, g.p1, g.p2) for g in d.gates])


def compare(got, want, blob_got, blob_want, what):
    for key in ("caps", "challenges", "openings", "fri"):
        g, w = np.asarray(got[key], np.uint64).reshape(-1), np.asarray(want[key], np.uint64).reshape(-1)
        assert g.shape == w.shape and (g == w).all(), "%s: HIP and oracle differ in %s (first at word %d)" % (
            what, key, int(np.nonzero(g != w)[0][0]) if g.shape == w.shape else -1)
    assert blob_got == blob_want, what + ": serialised proofs differ"


def flat(p):
    return np.concatenate([np.asarray(p[k], np.uint64).reshape(-1) for k in ("caps", "openings", "fri")])


def test_cyclic_step_circuit_at_paper_parameters_bit_exact_against_the_oracle():
    _cyclic_chained_step_parity(N, LOG_N, 192716)     # the proof size of the chain (DESIGN: the paper's "~200 kB", ivc_based_vpbs.rs:488)


def test_cyclic_step_circuit_n2048_bit_exact_against_the_oracle():
    """VERDICT r05 next 4 -- BASELINE config 5's ring, N = 2048 (src/ntt/params_2048.rs; pad to the next power of two as ivc_based_vpbs.rs:54-57 does
    by trial): 88 311 gate rows -> degree 2^17, LDE 2^20, 8 269 public inputs.  Until now this size was only VERIFIED (test_step_circuit_n2048,
    test_ivc_chain_tool[2048]); here one chained step is compared word for word with the C oracle: caps, challenges, openings, FRI words and all
    232 012 bytes -- the NTT shapes [4 4 | 4 4 1] / [4 4 | 4 4 4], the 2^20-leaf trees, the 2^17 partial products, quotient and FRI schedule
    (final polynomial 2^5 coefficients) inside one proof."""
    _cyclic_chained_step_parity(2048, 17, 232012)


def _cyclic_chained_step_parity(N, LOG_N, proof_bytes):
    """One CHAINED step of the IVC (at N = 1024: 46 656 gate rows of 12 gate types, degree 2^16, 4173 public inputs): base proof and step 0 on
    the GPU, then the wires of step 1 -- whose in-circuit verifier checks step 0's proof -- from the host witness plan; that one witness is
    proven by vpbs_prove_step and by the C oracle (its own FFTs, Merkle trees, partial products, gate constraints of the circuit's real gate
    set and selectors, quotient, openings, FRI), and the two proofs are the same words and the same bytes."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prove_ivc
    cyc_path, dum_path = export_circuits.ensure_cyclic_circuit(N, K, ELL, LOGB, N_LWE, LOG_N)
    ctx = vpbs_amd.Context(0, log_n_max=LOG_N)
    cyc, dum = prove_ivc.Circuit(ctx, cyc_path), prove_ivc.Circuit(ctx, dum_path)
    shape_words = cyc.d.meta["proof_words"]
    kn = K * N
    n_pi = len(cyc.d.pi_pos)
    keys = ctx.keygen(N, K, ELL, LOGB, N_LWE, 4242, 4.99027217501041e-8, 1.17021618159313e-5)
    testv, delta = api.testv(N, 2)
    ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta % P)
    acc_init = np.concatenate([np.zeros((K - 1, N), np.uint64), testv.reshape(1, N)])
    base_pis = np.concatenate([acc_init.reshape(-1), np.zeros(1 + kn + 8, np.uint64), cyc.vk])
    zero_pis = np.zeros(n_pi, np.uint64)

    def prove_dummy(pis):
        w = dum.plan.run(pis)
        d_w = torch.from_numpy(w.view(np.int64)).cuda()
        torch.cuda.synchronize()
        return dum.prove(d_w.data_ptr(), pis)[0]
    dummy_flat = flat(prove_dummy(zero_pis))          # the second proof slot (dummy_proof_and_vk)
    proof = prove_dummy(base_pis)                     # cyclic_base_proof (:292-299)
    pis_prev = base_pis
    steps = [(0, np.zeros(K * ELL * K * N, np.uint64), int(ct[N_LWE])), (1, keys["bsk"][0], int(ct[0]))]
    for s, (cond, ggsw, mask) in enumerate(steps):
        values = np.concatenate([flat(proof), pis_prev, np.array([cond], np.uint64), ggsw, np.array([mask], np.uint64), cyc.vk, dum.vk,
                                 dummy_flat, zero_pis])
        wires = cyc.plan.run(values)                  # one-shot host plan: the whole witness, in-circuit verifier rows included
        pis = wires[cyc.pi_cols, cyc.pi_rows].copy()
        d_w = torch.from_numpy(wires.view(np.int64)).cuda()
        torch.cuda.synchronize()
        proof, si = cyc.prove(d_w.data_ptr(), pis)
        pis_prev = pis
    assert int(pis[kn]) == 2                          # the counter: two chained steps
    ok, msg = cyc.d.circuit.check_witness(wires, api.hash_no_pad(pis))
    assert ok, msg
    # the same witness through the C oracle
    gs = oracle_gate_set(cyc.d)
    want = step_oracle.prove_step({"constants_sigmas": cyc.cs_values, "wires": wires, "quotient": None}, cyc.vk[:4], pis, LOG_N,
                                  sigmas=cyc.sigma, n_routed=80, n_constants=cyc.d.n_constants, gates=gs)
    assert (np.asarray(want["cs_cap"], np.uint64).reshape(-1) == np.asarray(cyc.cap, np.uint64).reshape(-1)).all()
    blob = ctx.step_proof_to_bytes(si, cyc.d.n_constants, proof)
    compare(proof, want, blob, step_oracle.to_bytes(want, want["ncols"], cyc.d.n_constants, pis, LOG_N), "cyclic circuit, chained step 1")
    assert len(blob) == proof_bytes
    assert cyc.verify(proof, pis) and step_oracle.verify_step(proof, want["cs_cap"], want["ncols"], cyc.vk[:4], pis, LOG_N)
    ctx.close()


def test_step_circuit_at_paper_parameters_bit_exact_against_the_oracle():
    """build_step_circuit (ivc_based_vpbs.rs:80-157, no recursive verifier) at N = 1024: 38 312 gate rows -> degree 2^16, 4105 public inputs; the
    exported sample witness proven by the HIP path and by the C oracle: identical words and bytes."""
    import torch
    d = circuit_file.load(export_circuits.ensure_step_circuit(N, K, ELL, LOGB, N_LWE))
    ctx = vpbs_amd.Context(0, log_n_max=d.log_n)
    sigma = d.circuit.sigma_values()
    cs_values = np.concatenate([d.constants, sigma])
    cs = ctx.commit_values(cs_values)
    digest = circuit_file.circuit_digest(cs.cap(), d.log_n)
    plan = d.circuit.witness_plan(d.preset_pos)
    wires = plan.run(d.sample_values)
    pi = np.array(d.pi_pos)
    pis = wires[pi[:, 0], pi[:, 1]].copy()
    assert (pis == d.sample_public_inputs).all()
    d_sigma = torch.from_numpy(sigma.view(np.int64)).cuda()
    d_w = torch.from_numpy(wires.view(np.int64)).cuda()
    torch.cuda.synchronize()
    si = ctx.make_step_inputs(d.log_n, d_w.data_ptr(), None, None, cs, digest, pis, on_device=True, shapes=(135, 20, 16),
                              sigmas=int(d_sigma.data_ptr()), n_routed=80, n_constants=d.n_constants, gates=d.gates)
    got = ctx.prove_step(si)
    want = step_oracle.prove_step({"constants_sigmas": cs_values, "wires": wires, "quotient": None}, digest, pis, d.log_n, sigmas=sigma,
                                  n_routed=80, n_constants=d.n_constants, gates=oracle_gate_set(d))
    compare(got, want, ctx.step_proof_to_bytes(si, d.n_constants, got), step_oracle.to_bytes(want, want["ncols"], d.n_constants, pis, d.log_n),
            "step circuit")
    plan.free()
    ctx.close()
