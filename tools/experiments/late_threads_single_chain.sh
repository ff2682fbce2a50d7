#!/bin/bash
# the late pool of ONE chain: 14 threads (two FRI queries per strand, the default of a 16-CPU host) against 20 and 28 (one query per strand; the
# pool then no longer fits one last-level-cache domain and is placed by the scheduler)
for rep in 1 2 3; do for t in 14 28 20; do
  VPBS_LATE_THREADS=$t VPBS_IVC_CHAINS=1 python tools/prove_ivc.py 1024 728 16 ${1:-300} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('late threads $t:', round(d['ms_per_step'],3), 'ms/step; late', round(s['witness_late_phase_host'],3), 'prove', round(s['prove_step'],3), 'load', round(d['host']['loadavg']))"
done; done
