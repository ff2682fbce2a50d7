one() { # cpus chains dw steps
  local pre=""; [ "$1" != "16" ] && pre="taskset -c 0-$(( $1 - 1 ))"
  VPBS_IVC_CHAINS=$2 VPBS_IVC_DEVICE_WITNESS=$3 timeout -k 5 300 $pre python tools/prove_ivc.py 1024 728 16 $4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('cpus=$1 chains=$2 dw=$3 steps=$4', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'late', round(s['witness_late_phase_host'],2), 'prove', round(s['prove_step'],2), 'load', round(d['host']['loadavg']))"
}
one 2 6 64 300; one 2 6 64 200; one 2 6 64 300; one 2 6 64 200; one 4 6 64 300; one 4 6 64 200; one 16 6 64 300; one 2 6 64 730
