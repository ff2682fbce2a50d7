#!/bin/bash
# Soak of the final kernels (round 5 wrote it; round 6 re-ran it after the spill / scalar-dispatch / queued-scatter changes) --  (residue arithmetic with rare wrap paths, radix-16 NTT, fused Poseidon groups): whole 730-step chains in the
# arrangements a deployment uses, every chain's last proof verified by vpbs_verify_pbs and decrypted by the tool, completion-word waits counted
# (VPBS_TRACE_SYNC).  One line per arrangement; the condensed record is profiles/rNN_soak.json.
# usage (GPU box): tools/experiments/soak_chains.sh [out_dir] [late]
out=${1:-gpurun_out/soak}; mkdir -p $out
run() { # name mask chains device_witness N log_degree
  local pre=""; [ -n "$2" ] && pre="taskset -c $2"
  VPBS_TRACE_SYNC=1 VPBS_IVC_CHAINS=$3 VPBS_IVC_DEVICE_WITNESS=$4 timeout -k 5 900 $pre python tools/prove_ivc.py ${5:-1024} 728 ${6:-16} 730 > $out/$1.json 2> $out/$1.err
  echo "$1 rc=$? $(grep -a '\[sync\]' $out/$1.err | tail -1) $(tail -1 $out/$1.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['chains'], 'chains', round(d['seconds'],1), 's', round(d['vpbs_proofs_per_s'],4), 'vPBS/s; decrypted', [d['decrypted']==d['message']]+[c['decrypted']==c['message'] for c in d['other_chains']], 'load', round(d['host']['loadavg']))")"
}
if [ "$2" = "late" ]; then   # only the arrangements with the LATE phase on the device too (round 6: the walk over eight workgroups)
  export VPBS_IVC_DEVICE_LATE=1
  run dw8_late_a "" 8 64
  run dw8_late_2cpus "0-1" 8 64
  run single_late "" 1 64
  run dw8_late_b "" 8 64
  exit 0
fi
run host8_a "" 8 0
run dw8_a "" 8 64
run dw8_2cpus "0-1" 8 64
run single_a "" 1 0
run host8_b "" 8 0
run dw8_4cpus "0-3" 8 64
run single_n2048 "" 1 0 2048 17
run host8_c "" 8 0
