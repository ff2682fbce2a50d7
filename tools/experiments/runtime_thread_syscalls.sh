#!/bin/bash
# What is the busy HSA runtime thread doing?  Samples /proc/PID/task/TID/{syscall,wchan} of the busiest thread that still carries the process name.
C=verifiable-fhe-paper_amd/circuits
VPBS_IVC_DEVICE_WITNESS=64 GPU_MAX_HW_QUEUES=8 taskset -c 0-1 ./examples/prove_ivc $C/cyclic_N1024_K2_ELL4_LOGB5_n728_deg16_slots2.bin $C/dummy_N1024_K2_ELL4_LOGB5_n728_deg16_slots2.bin 400 > /tmp/cxx_ivc.out 2>&1 &
pid=$!
sleep 4
tid=$(for t in /proc/$pid/task/*; do [ "$(basename $t)" != "$pid" ] && [ "$(cat $t/comm)" = "prove_ivc" ] && awk '{print $14+$15, $1}' $t/stat; done | sort -n -r | head -1 | awk '{print $2}')
echo "runtime thread $tid"
for i in $(seq 1 200); do
  echo "$(cat /proc/$pid/task/$tid/syscall 2>&1 | awk '{print $1, $2, $3}') | $(cat /proc/$pid/task/$tid/wchan 2>&1)"
  sleep 0.01
done | sort | uniq -c | sort -n -r | head -12
wait $pid
tail -1 /tmp/cxx_ivc.out | cut -c1-160
