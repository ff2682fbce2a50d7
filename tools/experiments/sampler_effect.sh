#!/bin/bash
# Does reading gpu_busy_percent every 50 ms next to a 2-CPU run change its speed?  (An accident of gpu_busy_sample.sh: 8.5 instead of 10.8 ms per proof.)
f=$(ls /sys/class/drm/card*/device/gpu_busy_percent 2>/dev/null | head -1)
run() { # sampler(0/1/2)
  VPBS_IVC_CHAINS=6 VPBS_IVC_DEVICE_WITNESS=64 taskset -c 0-1 python tools/prove_ivc.py 1024 728 16 250 > /tmp/run.out 2>/dev/null &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    [ "$1" = 1 ] && cat $f > /dev/null 2>&1
    [ "$1" = 2 ] && cat /proc/uptime > /dev/null
    sleep 0.05
  done
  wait $pid
  tail -1 /tmp/run.out | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sampler=$1', round(d['ms_per_step']/d['chains'],3), 'ms/proof', 'load', round(d['host']['loadavg']))"
}
for rep in 1 2; do run 1; run 0; run 2; done
