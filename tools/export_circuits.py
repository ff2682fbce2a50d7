#!/usr/bin/env python3
"""Produces the circuit FILES the prover-side tools load (verifiable-fhe-paper_amd/circuit_file.py only locates them): front end of
tools/export_step_circuit.py, the stand-in for the reference's Rust circuit builder (/root/reference/src/vtfhe/ivc_based_vpbs.rs:80-157
build_step_circuit, :159-275 the cyclic circuit).  Every export runs in a process of its own; nothing of the circuit builder is imported by
the caller.  __graft_entry__.build() calls ensure_standard(); tests call ensure_*() for the parameter sets they need (test infrastructure
may run the builder; bench.py, tools/prove_ivc.py and tools/prove_pbs.py may not and do not).

usage: tools/export_circuits.py                      the standard sets (below)
       tools/export_circuits.py --step N K ELL LOGB n_lwe
       tools/export_circuits.py --cyclic N K ELL LOGB n_lwe log_n"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from vpbs_amd import circuit_file  # noqa: E402

EXPORTER = os.path.join(ROOT, "tools", "export_step_circuit.py")
# what bench.py, tools/prove_ivc.py / prove_pbs.py and the examples load: the paper's parameters (main.rs:23-30; BASELINE configs 2-4), BASELINE
# config 5's ring (N = 2048 -> degree 2^17), and BASELINE config 1's ring (N = 8) for the smoke-sized chains
STANDARD_STEP = [(1024, 2, 4, 5, 728)]
STANDARD_CYCLIC = [(1024, 2, 4, 5, 728, 16), (2048, 2, 4, 5, 728, 17), (8, 2, 4, 5, 6, 13), (8, 2, 4, 5, 1, 13)]


def ensure_step_circuit(N=1024, K=2, ELL=4, LOGB=5, n_lwe=728):
    path = circuit_file.step_circuit_path(N, K, ELL, LOGB, n_lwe)
    if not os.path.exists(path):
        os.makedirs(circuit_file.DIR, exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        subprocess.check_call([sys.executable, EXPORTER, tmp] + [str(x) for x in (N, K, ELL, LOGB, n_lwe)], stdout=subprocess.DEVNULL)
        os.replace(tmp, path)
    return path


def ensure_cyclic_circuit(N=1024, K=2, ELL=4, LOGB=5, n_lwe=728, log_n=16):
    path, dummy = circuit_file.cyclic_circuit_paths(N, K, ELL, LOGB, n_lwe, log_n)
    if not (os.path.exists(path) and os.path.exists(dummy)):
        os.makedirs(circuit_file.DIR, exist_ok=True)
        tmp, tmpd = path + ".tmp%d" % os.getpid(), dummy + ".tmp%d" % os.getpid()
        subprocess.check_call([sys.executable, EXPORTER, "--cyclic", tmp, tmpd] + [str(x) for x in (N, K, ELL, LOGB, n_lwe, log_n)],
                              stdout=subprocess.DEVNULL)
        os.replace(tmpd, dummy)
        os.replace(tmp, path)
    return path, dummy


def ensure_standard():
    return [ensure_step_circuit(*a) for a in STANDARD_STEP] + [p for a in STANDARD_CYCLIC for p in ensure_cyclic_circuit(*a)]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--step":
        print(ensure_step_circuit(*[int(x) for x in sys.argv[2:7]]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--cyclic":
        print(*ensure_cyclic_circuit(*[int(x) for x in sys.argv[2:8]]))
    else:
        for p in ensure_standard():
            print(p)
