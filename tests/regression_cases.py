"""Shared by the CPU and GPU regression tests: rebuild the inputs of tests/golden/regression_step_proofs.json."""
import hashlib
import json
import os
import random

import numpy as np

import gates_oracle as go
from vpbs_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIGEST = [101, 202, 303, 404]
GATES = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing",
         "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"]


def cases():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "regression_step_proofs.json")))["cases"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def build(case):
    """-> dict(inputs, pis, log_n, sigma, n_constants, gate_spec) for one frozen case"""
    log_n = case["log_n"]
    if case["kind"] == "synthetic":
        return {"inputs": synth.step_inputs(log_n), "pis": synth.field_elements(case["pi_seed"], case["n_public_inputs"]), "log_n": log_n,
                "sigma": None, "n_constants": 0, "gates": None}
    rnd = random.Random(case["seed"])
    gs = go.GateSet(GATES)
    pis = [rnd.randrange(go.P) for _ in range(4)]
    constants, wires, sigma, _ = go.demo_circuit(rnd, gs, log_n, pis)
    return {"inputs": {"constants_sigmas": np.concatenate([constants, sigma]), "wires": wires, "quotient": None}, "pis": pis, "log_n": log_n,
            "sigma": sigma, "n_constants": constants.shape[0], "gates": GATES}


def check(case, proof):
    assert sha(proof["caps"]) == case["caps_sha256"], "caps changed"
    assert [int(x) for x in proof["challenges"]] == case["challenges"], "transcript challenges changed"
    assert sha(proof["openings"]) == case["openings_sha256"], "openings changed"
    assert proof["fri"].size == case["fri_words"] and int(proof["fri"][-1]) == case["pow_witness"], "FRI shape / pow witness changed"
    assert sha(proof["fri"]) == case["fri_sha256"], "FRI proof changed"
