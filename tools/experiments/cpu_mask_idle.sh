#!/bin/bash
# Are the CPUs of the taskset mask ours alone?  /proc/stat of cpu0 and cpu1 around a 2-CPU run: busy ticks of the two CPUs against the CPU time
# of the process itself (the box is shared: other tenants' threads run on the same hardware threads).
snap() { awk '/^cpu[01] /{busy+=$2+$3+$4+$7+$8; idle+=$5+$6} END{print busy, idle}' /proc/stat; }
read b0 i0 < <(snap)
t0=$(date +%s.%N)
VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=6 VPBS_IVC_DEVICE_WITNESS=64 taskset -c 0-1 python tools/prove_ivc.py 1024 728 16 200 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['cpu_by_role']; print('ms/proof', round(d['ms_per_step']/d['chains'],3), 'chain seconds', round(d['seconds'],2), 'cpu ms/proof', c['cpu_ms_per_chained_step'])"
t1=$(date +%s.%N)
read b1 i1 < <(snap)
awk -v b=$((b1-b0)) -v i=$((i1-i0)) -v w=$(echo "$t1 $t0" | awk '{print $1-$2}') 'BEGIN{printf "cpu0+cpu1 over the whole run (%.1f s wall): busy %.1f s, idle %.1f s\n", w, b/100, i/100}'
