#!/usr/bin/env python3
"""Gate rows of the reference's step circuit without its recursive verifier (circuitgen/step_circuit.py) for every ring dimension the
reference ships NTT parameters for, at the decomposition parameters of src/main.rs (k = 1, ELL = 4, LOGB = 5, n = 728).  CPU only."""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "circuitgen")]
import step_circuit as sc  # noqa: E402
from vpbs_amd import api  # noqa: E402

if __name__ == "__main__":
    print("| N | gate rows | Arithmetic | BaseSum | Poseidon | Constant | degree | public inputs | copy constraints |")
    print("|---|---|---|---|---|---|---|---|---|")
    for log_N in [int(a) for a in sys.argv[1:]] or range(3, 12):
        N = 1 << log_N
        t = time.time()
        circ = sc.StepCircuit(api, N, 2, 4, 5, 728, api.ntt_params(log_N))
        b = circ.built
        k = collections.Counter(b.row_kinds)
        print("| %d | %d | %d | %d | %d | %d | 2^%d | %d | %d |" % (N, b.used_rows, k["arithmetic"], k["base_sum"], k["poseidon"], k["constant"], b.log_n,
                                                             len(b.public_inputs), b.circuit.copies.shape[0]), flush=True)
