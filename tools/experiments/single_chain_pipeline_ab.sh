#!/bin/bash
# Run ON THE GPU BOX: ONE chain (the latency of one vPBS), 240 steps, alternating: early witness phases on the host (default) / on the device
# in batches of 64 (VPBS_IVC_DEVICE_WITNESS=64); ms per chained step and its split.  The host pipeline depends on the shared host's load.
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do for dw in 0 64; do
  VPBS_IVC_DEVICE_WITNESS=$dw python3 tools/prove_ivc.py 1024 728 16 ${1:-240} 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('device_witness=$dw ms_per_step %.2f' % d['ms_per_step'], {k: round(v,2) for k,v in d['ms_per_step_split'].items() if isinstance(v,(int,float))}, 'load', os.getloadavg()[0])"
done; done
