"""host-only verifier timing: vpbs_verify_step on a full-size (2^15) step proof produced on the GPU"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vpbs_amd
from vpbs_amd import api, synth
import bench
log_n = 15
ctx = vpbs_amd.Context(0, log_n_max=16)
gates = api.GateSet(bench.GATES)
inputs = synth.step_inputs(log_n, cols=bench.COLS)
cs = ctx.commit_values(inputs["constants_sigmas"])
pis = synth.field_elements(0xABCD, 77)
sig = np.ascontiguousarray(inputs["constants_sigmas"][bench.N_CONSTANTS:])
digest = np.array([11, 22, 33, 44], np.uint64)
si = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs, digest, pis, sigmas=sig, n_routed=80, n_constants=bench.N_CONSTANTS, gates=gates)
proof = ctx.prove_step(si)
cap = cs.cap()
ncols = [bench.COLS["constants_sigmas"], 135, 20, 16]
for check in (False, True):
    t = time.perf_counter()
    for _ in range(5):
        ok = api.verify_step(proof, cap, ncols, digest, pis, log_n, check_permutation=check, n_constants=bench.N_CONSTANTS, n_routed=80,
                             gates=gates if check else None)
    dt = (time.perf_counter() - t) / 5
    print("verify_step (FRI%s): %.2f ms, accepted=%s" % (" + vanishing identity with gates" if check else " only", dt * 1e3, ok))
