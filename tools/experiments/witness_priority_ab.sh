#!/bin/bash
# the device pipeline's witness contexts on streams of the least (-1) or most (1) urgent priority against default priority: needs a build with
# vpbs_ctx_create_with_priority (not kept: both lose 10 %)
one() { # prio cpus chains steps
  local pre=""; [ "$2" != "16" ] && pre="taskset -c 0-$(( $2 - 1 ))"
  VPBS_IVC_WITNESS_PRIORITY=$1 VPBS_IVC_CHAINS=$3 VPBS_IVC_DEVICE_WITNESS=64 timeout -k 5 400 $pre python tools/prove_ivc.py 1024 728 16 $4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('priority=$1 cpus=$2 chains=$3', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'load', round(d['host']['loadavg']))"
}
for rep in 1 2; do for p in 1 0; do one $p 16 8 200; one $p 4 8 200; done; done
