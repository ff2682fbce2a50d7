import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, vpbs_amd
from vpbs_amd import synth
import bench
log_n=15
ctx = vpbs_amd.Context(0, log_n_max=16)
gates = vpbs_amd.api.GateSet(bench.GATES)
inputs = synth.step_inputs(log_n, cols=bench.COLS)
dev = {k: torch.from_numpy(inputs[k].view(np.int64)).cuda() for k in ("wires","constants_sigmas")}
cs = ctx.commit_values(inputs["constants_sigmas"])
pis = synth.field_elements(0xABCD, 77)
sig_ptr = dev["constants_sigmas"].data_ptr() + 8*bench.N_CONSTANTS*(1<<log_n)
si = ctx.make_step_inputs(log_n, dev["wires"].data_ptr(), None, None, cs, np.array([11,22,33,44],np.uint64), pis, on_device=True, shapes=(135,20,16), sigmas=sig_ptr, n_routed=80, n_constants=bench.N_CONSTANTS, gates=gates)
for _ in range(3): ctx.prove_step(si)
print("TRACE_BEGIN", file=sys.stderr, flush=True)
ctx.prove_step(si)
print("TRACE_END", file=sys.stderr, flush=True)
