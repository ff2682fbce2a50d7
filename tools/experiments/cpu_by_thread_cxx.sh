#!/bin/bash
# CPU time by thread of the plain C++ host (examples/prove_ivc: no Python, no torch in the process) -- is the runtime thread that is busy
# whenever the device is busy the HIP / HSA runtime's, or PyTorch's?  user and system time apart: a spin in user space, or wake-ups through the kernel?
# usage (GPU box): [ENV=...] tools/experiments/cpu_by_thread_cxx.sh [steps=730]
steps=${1:-730}
C=verifiable-fhe-paper_amd/circuits
VPBS_IVC_DEVICE_WITNESS=64 GPU_MAX_HW_QUEUES=8 ./examples/prove_ivc $C/cyclic_N1024_K2_ELL4_LOGB5_n728_deg16_slots2.bin $C/dummy_N1024_K2_ELL4_LOGB5_n728_deg16_slots2.bin "$steps" > /tmp/cxx_ivc.out 2>&1 &
pid=$!
last=""
while kill -0 $pid 2>/dev/null; do
  snap=$(for t in /proc/$pid/task/*; do [ -r $t/stat ] && awk -v n="$(cat $t/comm 2>/dev/null)" -v sw="$(awk '/^voluntary_ctxt/{print $2}' $t/status 2>/dev/null)" '{print n, $1, $14, $15, sw}' $t/stat 2>/dev/null; done)
  case "$snap" in *vpbs-stager*) last="$snap"; at=$(awk '{print $1}' /proc/uptime);; esac
  [ -z "$t0" ] && t0=$(awk '{print $1}' /proc/uptime)
  sleep 0.2
done
wait $pid
tail -1 /tmp/cxx_ivc.out | cut -c1-200
echo "thread comm, tid, utime, stime (ticks of 10 ms), voluntary context switches at the last sample with the chain running, $(awk -v a="$at" -v b="$t0" "BEGIN{print a-b}") s after the start:"
echo "$last" | sort -k3 -n -r | head -${TOP:-4}
