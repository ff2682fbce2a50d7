#!/usr/bin/env python3
"""Writes the step circuit (tests/step_circuit.py: build_step_circuit of the reference without its recursive verifier) as the flat
circuit description a non-Python host hands to the C ABI -- the same arrays the Rust side would export from CircuitData after
builder.build(): gates, gate per row, constants columns, copy constraints, gadget generators, the targets the
PartialWitness sets, the public-input targets; plus one sample PartialWitness and the public inputs it must produce.
Format: little-endian u64 words, see examples/prove_step_circuit.cpp (the reader).
usage: tools/export_step_circuit.py OUT.bin [N K ELL LOGB n_lwe]     (default 8 2 4 5 6)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import step_circuit as sc  # noqa: E402
from vpbs_amd import api  # noqa: E402

MAGIC = 0x5354455043495243  # "STEPCIRC"


def export(path, N=8, K=2, ELL=4, LOGB=5, n_lwe=6, seed=1):
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n_lwe, api.ntt_params(N.bit_length() - 1))
    b = circ.built
    c = b.circuit
    n = b.n
    targets = ([t for p in circ.acc_init for t in p] + [t for p in circ.acc_in for t in p] + circ.ggsw_flat + [circ.counter, circ.mask] +
               circ.bsk_hash_in + circ.lwe_hash_in)
    pos = lambda t: b.pos(t)[0] * n + b.pos(t)[1]
    preset_pos = np.array([pos(t) for t in targets], np.uint64)
    pi_pos = np.array([pos(t) for t in b.public_inputs], np.uint64)
    rng = np.random.default_rng(seed)
    values = rng.integers(0, api.P, size=len(targets), dtype=np.uint64)
    values[len(targets) - 10] = 2                                   # counter: a CMUX step
    wires = c.generate_witness(dict(zip([b.pos(t) for t in targets], values)))
    pis = np.array(b.values(wires, b.public_inputs), np.uint64)
    gens = []
    for kind, p0, ins, outs in c.generator_list:
        gens += [api.GENERATOR_KINDS.index(kind), p0, len(ins), len(outs)] + [cc * n + rr for cc, rr in ins] + [cc * n + rr for cc, rr in outs]
    words = [np.array([MAGIC, b.log_n, c.n_wires, c.n_routed, b.gates.n, c.constants.shape[0], c.copies.shape[0], len(c.generator_list),
                       len(gens), preset_pos.size, pi_pos.size], np.uint64),
             np.array([[g.kind, g.p0, g.p1, g.p2] for g in b.gates], np.uint64).reshape(-1),
             c.row_gate.astype(np.uint64), c.constants.reshape(-1), c.copies.astype(np.uint64).reshape(-1), np.array(gens, np.uint64),
             preset_pos, pi_pos, values, pis, np.array([N, K, ELL, LOGB, n_lwe, b.used_rows], np.uint64)]
    with open(path, "wb") as f:
        for w in words:
            f.write(np.ascontiguousarray(w, dtype="<u8").tobytes())
    return circ, pis


if __name__ == "__main__":
    args = [int(x) for x in sys.argv[2:7]]
    circ, _ = export(sys.argv[1], *args)
    print("wrote %s: %d gate rows, degree 2^%d" % (sys.argv[1], circ.built.used_rows, circ.built.log_n))
