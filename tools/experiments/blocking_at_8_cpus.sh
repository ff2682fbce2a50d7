#!/bin/bash
# 8 and 10 CPUs, eight chains, device pipeline: AUTO (spin from 8 CPUs on) against sleeping waits
one() { # blocking cpus
  VPBS_CPU_BY_ROLE=1 VPBS_BLOCKING_SYNC=$1 VPBS_IVC_CHAINS=8 VPBS_IVC_DEVICE_WITNESS=64 timeout -k 5 400 taskset -c 0-$(( $2 - 1 )) python tools/prove_ivc.py 1024 728 16 200 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('blocking=$1 cpus=$2', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'cpu/proof', d['cpu_by_role']['cpu_ms_per_chained_step'], 'load', round(d['host']['loadavg']))"
}
for rep in 1 2; do one 0 8; one 1 8; one 0 10; one 1 10; done
