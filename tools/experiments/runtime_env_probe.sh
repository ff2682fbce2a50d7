run() { echo "== $*"; env "$@" VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=4 VPBS_IVC_DEVICE_WITNESS=64 taskset -c 0-1 python tools/prove_ivc.py 1024 728 16 150 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['cpu_by_role']; print(round(d['ms_per_step']/d['chains'],2), 'ms/proof', round(d['seconds'],2), 's;', c['cpu_ms_per_chained_step'], 'cpu ms/proof'); print(c['busiest_threads_cpu_s'][:6])"; }
run A=1
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=0
run AMD_DIRECT_DISPATCH=0
run HIP_FORCE_QUEUE_PROFILING=0 GPU_MAX_HW_QUEUES=4
