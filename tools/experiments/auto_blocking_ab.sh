one() { # env cpus chains dw steps
  local pre=""; [ "$2" != "16" ] && pre="taskset -c 0-$(( $2 - 1 ))"
  env $1 VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=$3 VPBS_IVC_DEVICE_WITNESS=$4 timeout -k 5 400 $pre python tools/prove_ivc.py 1024 728 16 $5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('$1 cpus=$2 chains=$3 dw=$4', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'late', round(s['witness_late_phase_host'],2), 'prove', round(s['prove_step'],2), 'cpu/proof', d['cpu_by_role']['cpu_ms_per_chained_step'], 'load', round(d['host']['loadavg']))"
}
for rep in 1 2; do
  one A=1 16 6 0 200; one VPBS_BLOCKING_SYNC=0 16 6 0 200
  one A=1 16 1 0 300; one VPBS_BLOCKING_SYNC=0 16 1 0 300
  one A=1 8 8 64 200
done
