one() { # chains dw steps
  VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=$1 VPBS_IVC_DEVICE_WITNESS=$2 timeout -k 5 400 python tools/prove_ivc.py 1024 728 16 $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('chains=$1 dw=$2', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'late', round(s['witness_late_phase_host'],2), 'prove', round(s['prove_step'],2), 'cpu/proof', d['cpu_by_role']['cpu_ms_per_chained_step'], 'load', round(d['host']['loadavg']))"
}
for rep in 1 2; do one 8 0 200; one 9 0 200; one 10 0 200; one 12 0 200; done
