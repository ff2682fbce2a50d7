one() { # queues cpus chains dw steps
  local pre=""; [ "$2" != "16" ] && pre="taskset -c 0-$(( $2 - 1 ))"
  GPU_MAX_HW_QUEUES=$1 VPBS_IVC_CHAINS=$3 VPBS_IVC_DEVICE_WITNESS=$4 timeout -k 5 300 $pre python tools/prove_ivc.py 1024 728 16 $5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('queues=$1 cpus=$2 chains=$3 dw=$4', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'late', round(s['witness_late_phase_host'],2), 'prove', round(s['prove_step'],2), 'load', round(d['host']['loadavg']))"
}
one 16 4 8 64 200; one 16 4 10 64 200; one 16 2 6 64 200; one 16 2 8 64 200; one 12 4 6 64 200; one 16 16 8 64 200; one 16 4 6 32 200
