echo "cpu0 siblings: $(cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list)  cpu1 siblings: $(cat /sys/devices/system/cpu/cpu1/topology/thread_siblings_list)  cpu2: $(cat /sys/devices/system/cpu/cpu2/topology/thread_siblings_list)"
grep -m1 "model name" /proc/cpuinfo; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
one() { # mask chains dw steps
  VPBS_IVC_CHAINS=$2 VPBS_IVC_DEVICE_WITNESS=$3 timeout -k 5 300 taskset -c $1 python tools/prove_ivc.py 1024 728 16 $4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('mask=$1 chains=$2', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'late', round(s['witness_late_phase_host'],2), 'prove', round(s['prove_step'],2), 'load', round(d['host']['loadavg']))"
}
one 0-1 6 64 250; one 0,2 6 64 250; one 8,24 6 64 250; one 0-1 6 64 250; one 0,2 6 64 250; one 8,24 6 64 250
