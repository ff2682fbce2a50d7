#!/bin/bash
# Run ON THE GPU BOX: eight host-pipeline chains with 1 / 2 / 4 / 8 threads in every chain's early-phase pool (round 5: the early phase runs
# AHEAD of the proof and with eight chains has 60 ms per step to finish in; its pool spins between its hundreds of levels) -- throughput and
# CPU time per chained proof by thread role.  -> gpurun_out/early_threads_<t>.json
cd "$GRAFT_REPO_ROOT"
for t in 8 2 1 4; do
  VPBS_EARLY_THREADS=$t VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=8 python3 tools/prove_ivc.py 1024 728 16 ${1:-200} > gpurun_out/early_threads_$t.json 2> gpurun_out/early_threads_$t.err
  python3 - <<P
import json
d=json.load(open("gpurun_out/early_threads_$t.json"))
c=d["cpu_by_role"]
print("early threads $t: %.2f ms per chained proof, %.1f CPU-ms per proof, by role %s" % (d["ms_per_step"]/8, c["cpu_ms_per_chained_step"], json.dumps(c["cpu_ms_per_chained_step_by_role"])))
P
done
