#!/bin/bash
# The late pool of ONE chain with 14 threads: confined to one last-level-cache domain (8 cores x 2 SMT threads on the EPYC 9575F: SMT siblings share
# a core) against placed by the scheduler (VPBS_POOL_PIN=0).  usage (GPU box): tools/experiments/pool_pin_ab.sh [steps=300]
steps=${1:-300}
for rep in 1 2 3; do
  for pin in 1 0; do
    VPBS_POOL_PIN=$pin VPBS_IVC_CHAINS=1 python tools/prove_ivc.py 1024 728 16 "$steps" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('pin=$pin', round(d['ms_per_step'],3), 'ms/step; late', round(s['witness_late_phase_host'],3), 'prove', round(s['prove_step'],3), 'loadavg', round(d['host']['loadavg']))"
  done
done
