"""Randomised stress of the gate-constraint kernels against oracle/gates.c: random gate subsets and parameters, random columns,
edge-valued columns (0, 1, p-1, 2^32-1, 2^63), 1..4 challenges.  Run on the GPU box: python tools/stress_gates.py [cases]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen")]
import numpy as np, torch
import gates_oracle as go
import vpbs_amd
from vpbs_amd import api
P = api.P
POOL = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing", "reducing_ext",
        ("random_access", 4), "exponentiation", "coset_interpolation", ("base_sum", 10, 3), ("base_sum", 31, 4), ("random_access", 1), ("random_access", 2),
        ("random_access", 3), ("random_access", 5), ("coset_interpolation", 2), ("coset_interpolation", 3), ("coset_interpolation", 5), ("arithmetic", 3),
        ("constant", 1), ("reducing", 5), ("reducing_ext", 1), ("exponentiation", 7), ("mul_ext", 2), ("arithmetic_ext", 1)]
EDGE = np.array([0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 1 << 63, (1 << 32) - 2], dtype=np.uint64)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    r = random.Random(12345)
    rng = np.random.default_rng(999)
    ctx = vpbs_amd.Context(0, log_n_max=16)
    for case in range(cases):
        k = r.randint(1, 9)
        spec, seen = [], set()
        for s in r.sample(POOL, len(POOL)):
            name = s if isinstance(s, str) else s[0]
            if name not in seen:
                seen.add(name)
                spec.append(s)
            if len(spec) == k:
                break
        try:
            gs, ps = go.GateSet(spec), api.GateSet(spec)
        except (AssertionError, api.VpbsError):
            continue   # a degree-8 gate in a multi-selector set is rejected by both
        log_n = r.randint(3, 9)
        n = 1 << log_n
        nc = r.randint(1, 4)
        n_const = gs.num_selectors + gs.num_constants
        consts = rng.integers(0, P, size=(n_const + 1, n), dtype=np.uint64)
        wires = rng.integers(0, P, size=(135, n), dtype=np.uint64)
        if case % 3 == 0:   # sprinkle edge values
            m = rng.random(wires.shape) < 0.3
            wires[m] = EDGE[rng.integers(0, EDGE.size, size=int(m.sum()))]
            m = rng.random(consts.shape) < 0.3
            consts[m] = EDGE[rng.integers(0, EDGE.size, size=int(m.sum()))]
        pi_hash = [int(x) for x in rng.integers(0, P, size=4, dtype=np.uint64)]
        alphas = [int(x) for x in rng.integers(0, P, size=nc, dtype=np.uint64)]
        if case % 5 == 0:
            alphas[0] = P - 1
        cs, wb = ctx.commit_values(consts), ctx.commit_values(wires)
        out = torch.zeros((nc, 8 * n), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.set_gate_lanes(3 if case % 2 else 1)
        ctx.gate_terms(cs, wb, ps, pi_hash, alphas, out.data_ptr())
        ctx.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        idx = np.array([int(format(t, "0%db" % (log_n + 3))[::-1], 2) for t in range(8 * n)])
        got = got[:, idx]
        want = gs.terms_coset(cs.coeffs()[:n_const], wb.coeffs(), pi_hash, alphas)
        assert (got == want).all(), (case, spec, log_n, nc)
        cs.free(); wb.free()
        print("case %d ok: %d gates, log_n %d, nc %d" % (case, len(spec), log_n, nc), flush=True)
    ctx.close()
    print("STRESS_OK")


if __name__ == "__main__":
    main()
