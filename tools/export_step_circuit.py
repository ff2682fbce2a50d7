#!/usr/bin/env python3
"""Writes the step circuit (circuitgen/step_circuit.py: build_step_circuit of the reference without its recursive verifier) as the flat
circuit description a non-Python host hands to the C ABI -- the same arrays the Rust side would export from CircuitData after
builder.build(): gates, gate per row, constants columns, copy constraints, gadget generators, the targets the
PartialWitness sets, the public-input targets; plus one sample PartialWitness and the public inputs it must produce.
Format: little-endian u64 words, see examples/prove_step_circuit.cpp (the reader).
usage: tools/export_step_circuit.py OUT.bin [N K ELL LOGB n_lwe]     (default 8 2 4 5 6)
       tools/export_step_circuit.py --cyclic OUT.bin DUMMY.bin N K ELL LOGB n_lwe log_n"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "circuitgen")]
import step_circuit as sc  # noqa: E402
from vpbs_amd import api  # noqa: E402

MAGIC = 0x5354455043495243  # "STEPCIRC"
KIND_STEP, KIND_CYCLIC, KIND_DUMMY = 0, 1, 2


def write_circuit(path, built, preset_pos, values, pis, trailer):
    """the flat description of one built circuit (format: verifiable-fhe-paper_amd/circuit_file.py)"""
    c = built.circuit
    n = built.n
    pos = lambda cr: cr[0] * n + cr[1]
    pi_pos = np.array([pos(built.pos(t)) for t in built.public_inputs], np.uint64)
    gens = []
    for kind, p0, ins, outs in c.generator_list:
        gens += [api.GENERATOR_KINDS.index(kind), p0, len(ins), len(outs)] + [cc * n + rr for cc, rr in ins] + [cc * n + rr for cc, rr in outs]
    preset_pos = np.array([pos(p) for p in preset_pos], np.uint64)
    words = [np.array([MAGIC, built.log_n, c.n_wires, c.n_routed, built.gates.n, c.constants.shape[0], c.copies.shape[0], len(c.generator_list),
                       len(gens), preset_pos.size, pi_pos.size], np.uint64),
             np.array([[g.kind, g.p0, g.p1, g.p2] for g in built.gates], np.uint64).reshape(-1),
             c.row_gate.astype(np.uint64), c.constants.reshape(-1), c.copies.astype(np.uint64).reshape(-1), np.array(gens, np.uint64),
             preset_pos, pi_pos, np.asarray(values, np.uint64), np.asarray(pis, np.uint64), np.array(trailer, np.uint64)]
    with open(path, "wb") as f:
        for w in words:
            f.write(np.ascontiguousarray(w, dtype="<u8").tobytes())


def export(path, N=8, K=2, ELL=4, LOGB=5, n_lwe=6, seed=1):
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n_lwe, api.ntt_params(N.bit_length() - 1))
    b = circ.built
    targets = ([t for p in circ.acc_init for t in p] + [t for p in circ.acc_in for t in p] + circ.ggsw_flat + [circ.counter, circ.mask] +
               circ.bsk_hash_in + circ.lwe_hash_in)
    rng = np.random.default_rng(seed)
    values = rng.integers(0, api.P, size=len(targets), dtype=np.uint64)
    values[len(targets) - 10] = 2                                   # counter: a CMUX step
    wires = b.circuit.generate_witness(dict(zip([b.pos(t) for t in targets], values)))
    pis = np.array(b.values(wires, b.public_inputs), np.uint64)
    write_circuit(path, b, [b.pos(t) for t in targets], values, pis, [N, K, ELL, LOGB, n_lwe, b.used_rows])
    return circ, pis


def export_cyclic(path, dummy_path, N, K, ELL, LOGB, n_lwe, log_n):
    """the CYCLIC step circuit (circuitgen/cyclic_circuit.py: build_step_circuit + the in-circuit verifier of its own previous proof,
    ivc_based_vpbs.rs:159-275) and the dummy circuit of its base case.  PartialWitness order of the cyclic file: the inner proof's words
    (caps, openings, FriProof as vpbs_prove_step emits them), the inner proof's public inputs, the condition bit, the GGSW, the mask, the
    circuit's own verifier data (digest, cap), the dummy circuit's verifier data, the dummy circuit's proof (second proof slot) and its public
    inputs.  No sample witness (it would need a proof): the sample
    sections are zero and the trailer says so (kind)."""
    import cyclic_circuit as cyc
    cy = cyc.CyclicStepCircuit(api, N, K, ELL, LOGB, n_lwe, api.ntt_params(N.bit_length() - 1), log_n)
    b = cy.built
    write_circuit(path, b, cy.positions, np.zeros(len(cy.positions), np.uint64), np.zeros(len(b.public_inputs), np.uint64),
                  [N, K, ELL, LOGB, n_lwe, b.used_rows, KIND_CYCLIC, cy.shape.proof_words])
    dm = cyc.DummyCircuit(api, log_n, cy.shape.n_pi)
    d = dm.built
    write_circuit(dummy_path, d, [d.pos(t) for t in dm.pis], np.zeros(len(dm.pis), np.uint64), np.zeros(len(dm.pis), np.uint64),
                  [N, K, ELL, LOGB, n_lwe, d.used_rows, KIND_DUMMY, 0])
    return cy, dm


if __name__ == "__main__":
    if sys.argv[1] == "--cyclic":   # --cyclic OUT.bin DUMMY.bin N K ELL LOGB n_lwe log_n
        a = [int(x) for x in sys.argv[4:10]]
        cy, dm = export_cyclic(sys.argv[2], sys.argv[3], *a)
        print("wrote %s: %d gate rows, degree 2^%d, %d public inputs; %s: dummy circuit" % (sys.argv[2], cy.built.used_rows, cy.built.log_n,
                                                                                          cy.shape.n_pi, sys.argv[3]))
    else:
        args = [int(x) for x in sys.argv[2:7]]
        circ, _ = export(sys.argv[1], *args)
        print("wrote %s: %d gate rows, degree 2^%d" % (sys.argv[1], circ.built.used_rows, circ.built.log_n))
