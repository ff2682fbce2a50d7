#!/usr/bin/env python3
"""Per-gate cost of the gate-constraint kernels on the GPU box: every gate type alone, the 14 standard gates together (bench.py's synthetic
step) and the gate set of the cyclic step circuit, at degree 2^16 (LDE 2^19 points, 2 challenges), HIP-event time of the `gate_constraints`
group.  usage: python tools/time_gates.py [log_n=16] [reps=10]  ->  one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch

import vpbs_amd
from vpbs_amd import api

ALL = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing",
       "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"]
CYCLIC = [g for g in ALL if g != "exponentiation"]


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n = 1 << log_n
    ctx = vpbs_amd.Context(0, log_n_max=max(16, log_n))
    rng = np.random.default_rng(5)
    wires = rng.integers(0, api.P, size=(135, n), dtype=np.uint64)
    wb = ctx.commit_values(wires)
    out = torch.zeros((2, 8 * n), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    pi_hash = [1, 2, 3, 4]
    alphas = [int(x) for x in rng.integers(0, api.P, size=2, dtype=np.uint64)]
    res = {}
    sets = [(g if isinstance(g, str) else g[0], ["noop", g]) for g in ALL[1:]] + [("all_14", ALL), ("cyclic_13", CYCLIC)]
    for lanes, tile in ((1, 1), (3, 1), (1, 0)):
        ctx.set_gate_lanes(lanes)
        ctx.set_option("gates_tile", tile)
        for name, spec in sets:
            if (lanes == 3 or tile == 0) and not name.startswith(("all", "cyclic")):
                continue
            ps = api.GateSet(spec)
            consts = rng.integers(0, api.P, size=(ps.num_selectors + ps.num_constants, n), dtype=np.uint64)
            cs = ctx.commit_values(consts)
            ctx.gate_terms(cs, wb, ps, pi_hash, alphas, out.data_ptr())
            ctx.synchronize()
            ctx.timing_enable(1)
            ctx.timing_report()
            for _ in range(reps):
                ctx.gate_terms(cs, wb, ps, pi_hash, alphas, out.data_ptr())
            ctx.synchronize()
            rep = ctx.timing_report()
            ctx.timing_enable(0)
            t = sum(v["ms"] for k, v in rep.items()) / reps
            res["%s%s%s" % (name, "" if lanes == 1 else "_3lanes", "" if tile else "_tile_x_item_kernel")] = round(1e3 * t, 1)
            cs.free()
    print(json.dumps({"log_n": log_n, "points": 8 * n, "us_per_call": res}))


if __name__ == "__main__":
    main()
