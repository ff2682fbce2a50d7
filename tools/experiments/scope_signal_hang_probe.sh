#!/bin/bash
# Which configuration hangs with ROC_SYSTEM_SCOPE_SIGNAL=0?  Every run under its own timeout; SIGABRT on expiry makes faulthandler print the Python stacks.
run() {   # name taskset chains device_witness
  local pre=""
  [ -n "$2" ] && pre="taskset -c $2"
  echo "=== $1"
  ROC_SYSTEM_SCOPE_SIGNAL=0 VPBS_IVC_CHAINS=$3 VPBS_IVC_DEVICE_WITNESS=$4 timeout -s ABRT -k 5 90 $pre python -X faulthandler tools/prove_ivc.py 1024 728 16 100 > /tmp/probe.out 2>&1
  echo "rc=$?"; tail -c 1500 /tmp/probe.out | grep -v "^$" | tail -25 | cut -c1-300
}
run chain1_dw "" 1 64
run chains6_dw "" 6 64
run chains6_host "" 6 0
run chains6_dw_2cpus 0-1 6 64
