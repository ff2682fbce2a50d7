#!/bin/bash
# gpu_busy_percent of the device, sampled while six chains run with 2 CPUs and with all of them: is the device short of work at 2 CPUs?
f=$(ls /sys/class/drm/card*/device/gpu_busy_percent 2>/dev/null | head -1)
echo "sysfs: $f"
for cpus in 2 16; do
  pre=""; [ $cpus != 16 ] && pre="taskset -c 0-$((cpus-1))"
  VPBS_IVC_CHAINS=6 VPBS_IVC_DEVICE_WITNESS=64 $pre python tools/prove_ivc.py 1024 728 16 300 > /tmp/run.out 2>/dev/null &
  pid=$!
  : > /tmp/busy.txt
  while kill -0 $pid 2>/dev/null; do cat $f >> /tmp/busy.txt 2>/dev/null; sleep 0.05; done
  wait $pid
  tail -1 /tmp/run.out | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cpus=$cpus', round(d['ms_per_step']/d['chains'],3), 'ms/proof; chain phase', round(d['seconds'],1), 's')"
  python - <<PY
v=[int(x) for x in open('/tmp/busy.txt').read().split()]
n=len(v); tail=v[n//2:]   # the second half of the run is inside the chains
print('samples', n, 'busy % (second half of the run): mean', round(sum(tail)/max(1,len(tail)),1), 'min', min(tail), 'share of samples below 90:', round(sum(1 for x in tail if x<90)/len(tail),2))
PY
done
