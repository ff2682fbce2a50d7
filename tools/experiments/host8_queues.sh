one() { # env chains steps
  env $1 VPBS_IVC_CHAINS=$2 VPBS_IVC_DEVICE_WITNESS=0 timeout -k 5 400 python tools/prove_ivc.py 1024 728 16 $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('$1 chains=$2', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'late', round(s['witness_late_phase_host'],2), 'load', round(d['host']['loadavg']))"
}
for rep in 1 2; do one GPU_MAX_HW_QUEUES=8 8 200; one GPU_MAX_HW_QUEUES=12 8 200; one GPU_MAX_HW_QUEUES=16 8 200; one VPBS_LATE_THREADS=6 8 200; done
