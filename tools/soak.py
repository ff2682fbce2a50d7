"""soak: many step proofs in a row on one context -- identical proofs, flat device-memory use, stable time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vpbs_amd
from vpbs_amd import api, synth
import bench
log_n, N = bench.LOG_N, int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = vpbs_amd.Context(0, log_n_max=16)
gates = api.GateSet(bench.GATES)
inputs = synth.step_inputs(log_n, cols=bench.COLS)
dev = {k: torch.from_numpy(inputs[k].view(np.int64)).cuda() for k in ("wires", "constants_sigmas")}
cs = ctx.commit_values(inputs["constants_sigmas"])
pis = synth.field_elements(0xABCD, 77)
sig_ptr = dev["constants_sigmas"].data_ptr() + 8 * bench.N_CONSTANTS * (1 << log_n)
si = ctx.make_step_inputs(log_n, dev["wires"].data_ptr(), None, None, cs, np.array([11, 22, 33, 44], np.uint64), pis, on_device=True,
                          shapes=(135, 20, 16), sigmas=sig_ptr, n_routed=80, n_constants=bench.N_CONSTANTS, gates=gates)
first = ctx.prove_step(si)
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
times = []
for i in range(N):
    t = time.perf_counter()
    p = ctx.prove_step(si)
    times.append(time.perf_counter() - t)
    if i % 50 == 0:
        assert (p["fri"] == first["fri"]).all() and (p["caps"] == first["caps"]).all()
free1 = torch.cuda.mem_get_info()[0]
times = np.array(times) * 1e3
print("steps %d: median %.3f ms, p99 %.3f ms, max %.3f ms; free HBM before/after %.1f / %.1f MiB" %
      (N, np.median(times), np.percentile(times, 99), times.max(), free0 / 2**20, free1 / 2**20))
