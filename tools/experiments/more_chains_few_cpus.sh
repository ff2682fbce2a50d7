one() { # cpus chains dw steps
  local pre=""; [ "$1" != "16" ] && pre="taskset -c 0-$(( $1 - 1 ))"
  VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=$2 VPBS_IVC_DEVICE_WITNESS=$3 timeout -k 5 400 $pre python tools/prove_ivc.py 1024 728 16 $4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; c=d['cpu_by_role']; print('cpus=$1 chains=$2 dw=$3', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'late', round(s['witness_late_phase_host'],2), 'prove', round(s['prove_step'],2), 'cpu/proof', c['cpu_ms_per_chained_step'], 'load', round(d['host']['loadavg']))"
}
one 2 10 64 200; one 2 12 64 200; one 2 8 64 200; one 4 10 64 200; one 4 12 64 200; one 2 10 32 200
