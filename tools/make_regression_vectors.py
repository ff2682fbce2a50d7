#!/usr/bin/env python3
"""Freeze the oracle's outputs for a few seeded step proofs: tests/golden/regression_step_proofs.json.

These are REGRESSION vectors produced by this repository's own CPU oracle -- not golden vectors of the reference (the reference cannot
be run here; parity with real plonky2 stays unpinned).  They catch an accidental change that moves the oracle and the product together.
usage: python tools/make_regression_vectors.py   (rewrites the file)"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen")]
import numpy as np  # noqa: E402

import gates_oracle as go  # noqa: E402
import step_oracle  # noqa: E402
from vpbs_amd import synth  # noqa: E402

GATES = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing",
         "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"]
DIGEST = [101, 202, 303, 404]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def case_synthetic(log_n):
    """random columns, supplied Z/partial products and quotient chunks (the round-1 first-bar step)"""
    inputs = synth.step_inputs(log_n)
    pis = synth.field_elements(0x600D + log_n, 9)
    p = step_oracle.prove_step(inputs, DIGEST, pis, log_n)
    return {"kind": "synthetic", "log_n": log_n, "n_public_inputs": 9, "pi_seed": 0x600D + log_n}, p


def case_circuit(log_n, seed):
    """the 14-gate demo circuit: partial products, gate constraints and quotient computed by the prover"""
    rnd = random.Random(seed)
    gs = go.GateSet(GATES)
    pis = [rnd.randrange(go.P) for _ in range(4)]
    constants, wires, sigma, _ = go.demo_circuit(rnd, gs, log_n, pis)
    inputs = {"constants_sigmas": np.concatenate([constants, sigma]), "wires": wires, "quotient": None}
    p = step_oracle.prove_step(inputs, DIGEST, pis, log_n, sigmas=sigma, n_routed=80, n_constants=constants.shape[0], gates=gs)
    return {"kind": "circuit", "log_n": log_n, "seed": seed}, p


def case_bench():
    """the instance bench.py times (tests/regression_cases.py BENCH): degree 2^16, 135/20/16/86 columns, 14 gate types, 4173 public inputs"""
    import hashlib
    import regression_cases as rc
    b = rc.build({"kind": "bench", "log_n": rc.BENCH["log_n"]})
    p = step_oracle.prove_step(b["inputs"], b["digest"], b["pis"], b["log_n"], sigmas=b["sigma"], n_routed=80, n_constants=b["n_constants"],
                               gates=go.GateSet(GATES))
    blob = step_oracle.to_bytes(p, p["ncols"], b["n_constants"], b["pis"], b["log_n"])
    return {"kind": "bench", "log_n": b["log_n"], "cs_cap_sha256": sha(p["cs_cap"]), "bytes_sha256": hashlib.sha256(blob).hexdigest(),
            "bytes_len": len(blob)}, p


def main():
    out = []
    for meta, p in [case_synthetic(5), case_synthetic(8), case_circuit(6, 4242), case_circuit(7, 777), case_bench()]:
        meta.update({"caps_sha256": sha(p["caps"]), "openings_sha256": sha(p["openings"]), "fri_sha256": sha(p["fri"]),
                     "challenges": [int(x) for x in p["challenges"]], "pow_witness": int(p["fri"][-1]), "fri_words": int(p["fri"].size)})
        out.append(meta)
    path = os.path.join(ROOT, "tests", "golden", "regression_step_proofs.json")
    json.dump({"note": "REGRESSION vectors from this repository's CPU oracle (tools/make_regression_vectors.py); not reference golden vectors",
               "cases": out}, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
