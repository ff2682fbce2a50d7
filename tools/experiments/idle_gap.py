#!/usr/bin/env python3
"""Does a step proof run slower when the GPU idles between proofs (as it does in the IVC chain while the host generates the late witness)?
Proves the synthetic full-size step back to back and with a host-side pause before every proof; prints per-kernel-group device times and
the shader clock (vpbs_k_clock_probe) right before and right after each proof.
Measured (round 2, MI355X): 8.5 ms back to back; 8.9 / 9.2 / 9.4 ms after pauses of 1 / 2.5 / 5 ms -- every kernel group 7-14 % slower --
while the shader clock of a LIGHT probe kernel reads 2.40-2.43 GHz in all cases.  Under rocprofv3 --pmc GRBM_GUI_ACTIVE
(tools/experiments/gap_pmc.sh) the leaf-hash kernel takes the same number of cycles everywhere (15.8 M per XCD-sum) but runs at an
effective 2.15 GHz back to back and 2.0 GHz in the runs with pauses: the power management brings a loaded chip back to its sustained
clock slowly after every pause.  A spin kernel that kept every CU busy during the pause (integer arithmetic, with and
without a stream of memory reads, 1 to 8 workgroups per CU) changed nothing, so it was dropped again; real work does (three IVC chains side by
side: 8.5 ms of device time per proof)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vpbs_amd  # noqa: E402
from vpbs_amd import synth  # noqa: E402

log_n = 16
ctx = vpbs_amd.Context(0, log_n_max=16)
inputs = synth.step_inputs(log_n)
digest = np.array([1, 2, 3, 4], np.uint64)
pis = synth.field_elements(99, 40)
cs = ctx.commit_values(inputs["constants_sigmas"])
d = {k: torch.from_numpy(inputs[k].view(np.int64)).cuda() for k in ("wires", "zs_partial_products", "quotient")}
torch.cuda.synchronize()
si = ctx.make_step_inputs(log_n, d["wires"].data_ptr(), d["zs_partial_products"].data_ptr(), d["quotient"].data_ptr(), cs, digest, pis, on_device=True,
                          shapes=(135, 20, 16))
for _ in range(5):
    ctx.prove_step(si)
for gap_ms in (0.0, 1.0, 2.5, 5.0, 10.0, 0.0):
    ctx.timing_enable(1)
    ctx.timing_report()
    t_prove = 0.0
    n = 40
    before, after = [], []
    for _ in range(n):
        if gap_ms:
            t_end = time.perf_counter() + gap_ms * 1e-3
            while time.perf_counter() < t_end:
                pass
        before.append(ctx.clock_probe())
        t = time.perf_counter()
        ctx.prove_step(si)
        t_prove += time.perf_counter() - t
        after.append(ctx.clock_probe())
    rep = ctx.timing_report()
    under_load = ctx.timing_shader_clock()[0]
    ctx.timing_enable(0)
    print("gap %.1f ms: prove %.3f ms; leaf_hash %.3f coset_lde %.3f merkle_levels %.3f fri_tree %.3f (ms per step); shader clock before / after a proof %.0f / %.0f MHz, inside the leaf-hash kernel %.0f MHz" % (
        gap_ms, 1e3 * t_prove / n, *(rep[k]["ms"] / n for k in ("leaf_hash", "coset_lde", "merkle_levels", "fri_tree")), sum(before) / n, sum(after) / n, under_load))
