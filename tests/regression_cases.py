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


# the instance bench.py times and compares (BASELINE config 2 at the degree the reference builds for N = 1024): 2^16 rows, 135 wire /
# 20 Z+partial-product / 16 quotient / 86 constant+sigma columns of seeded field elements, all 14 gate types, 4173 public inputs
BENCH = {"log_n": 16, "cols": {"constants_sigmas": 86, "wires": 135, "zs_partial_products": 20, "quotient": 16}, "n_constants": 6,
         "n_routed": 80, "n_public_inputs": 4173, "pi_seed": 0xABCD, "digest": [11, 22, 33, 44]}


def cases(full_size=False):
    """full_size: the bench.py instance too (the oracle needs ~10-40 s of all host cores for it: GPU suite only)"""
    cs = json.load(open(os.path.join(ROOT, "tests", "golden", "regression_step_proofs.json")))["cases"]
    return [c for c in cs if full_size or c["kind"] != "bench"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def build(case):
    """-> dict(inputs, pis, log_n, sigma, n_constants, gate_spec) for one frozen case"""
    log_n = case["log_n"]
    if case["kind"] == "synthetic":
        return {"inputs": synth.step_inputs(log_n), "pis": synth.field_elements(case["pi_seed"], case["n_public_inputs"]), "log_n": log_n,
                "sigma": None, "n_constants": 0, "gates": None}
    if case["kind"] == "bench":
        inputs = synth.step_inputs(log_n, cols=BENCH["cols"])
        nc, nr = BENCH["n_constants"], BENCH["n_routed"]
        inputs["quotient"] = None
        return {"inputs": inputs, "pis": synth.field_elements(BENCH["pi_seed"], BENCH["n_public_inputs"]), "log_n": log_n,
                "sigma": np.ascontiguousarray(inputs["constants_sigmas"][nc:nc + nr]), "n_constants": nc, "gates": GATES, "digest": BENCH["digest"]}
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


def check_bytes(case, blob):
    if "bytes_sha256" in case:
        assert hashlib.sha256(blob).hexdigest() == case["bytes_sha256"] and len(blob) == case["bytes_len"], "serialised proof changed"
