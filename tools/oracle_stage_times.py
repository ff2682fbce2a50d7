"""stage timing of the CPU oracle's step proof (the cpu_baseline of bench.py) on this machine's host cores"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen")]
import numpy as np, oracle as orc, step_oracle, gates_oracle as go, pymodel
from vpbs_amd import synth
import bench
log_n = 15
inputs = synth.step_inputs(log_n, cols=bench.COLS)
T = {}
def timed(name, f):
    t = time.time(); r = f(); T[name] = T.get(name, 0) + time.time() - t; return r
cs = orc.Batch(inputs["constants_sigmas"], 3, 4, True)
sig = np.ascontiguousarray(inputs["constants_sigmas"][bench.N_CONSTANTS:])
gs = go.GateSet(bench.GATES)
pis = synth.field_elements(0xABCD, 77)
t_all = time.time()
w = timed("wires commit", lambda: orc.Batch(inputs["wires"], 3, 4, True))
ch = orc.ChallengerState(); ch.observe([11, 22, 33, 44]); ch.observe(orc.hash_no_pad(pis)); ch.observe(w.cap())
betas, gammas = ch.get_n(2), ch.get_n(2)
zs = timed("partial products", lambda: orc.partial_products(inputs["wires"][:80], sig, betas, gammas))
zb = timed("zs commit", lambda: orc.Batch(zs, 3, 4, True))
ch.observe(zb.cap()); alphas = ch.get_n(2)
gt = timed("gate constraints", lambda: gs.terms_coset(cs.coeffs()[:bench.N_CONSTANTS], w.coeffs(), orc.hash_no_pad(pis), alphas))
q = timed("quotient", lambda: orc.quotient_permutation(w.coeffs()[:80], cs.coeffs()[bench.N_CONSTANTS:], zb.coeffs(), betas, gammas, alphas, gate_terms=gt))
qb = timed("quotient commit", lambda: orc.Batch(q, 3, 4, False))
ch.observe(qb.cap()); zeta = ch.get_ext()
oracles = [cs, w, zb, qb]
ncols = [o.ncols for o in oracles]
batches, zeta_next = step_oracle.step_batches(ncols, 2, zeta, log_n)
openings = timed("openings", lambda: np.concatenate([o.eval_ext(zeta) for o in oracles] + [zb.eval_ext(zeta_next)[:2]]))
ch.observe(openings)
fri = timed("fri (prove_openings)", lambda: orc.prove_openings(oracles, batches, ch, orc.fri_params(log_n), log_n))
print("total %.2f s with %d OpenMP threads (os.cpu_count() = %d)" % (time.time() - t_all, orc.effective_cpus(), os.cpu_count()))
for k, v in T.items():
    print("  %-22s %.3f s" % (k, v))
