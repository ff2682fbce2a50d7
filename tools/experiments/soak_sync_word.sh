#!/bin/bash
# Soak of the completion-word waits: whole 730-step chains in the arrangements a deployment uses; every chain's last proof is verified and
# decrypted by the tool; VPBS_TRACE_SYNC prints at exit how many waits there were and how many the 200 ms runtime check had to end (expected 0).
out=${1:-gpurun_out/soak_word}; mkdir -p $out
run() { # name mask chains dw
  local pre=""; [ -n "$2" ] && pre="taskset -c $2"
  VPBS_TRACE_SYNC=1 VPBS_IVC_CHAINS=$3 VPBS_IVC_DEVICE_WITNESS=$4 timeout -k 5 900 $pre python tools/prove_ivc.py 1024 728 16 730 > $out/$1.json 2> $out/$1.err
  echo "$1 rc=$? $(grep -a '\[sync\]' $out/$1.err | tail -1) $(tail -1 $out/$1.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['chains'], 'chains', round(d['seconds'],1), 's', round(d['vpbs_proofs_per_s'],4), 'vPBS/s; decrypted', [d['decrypted']==d['message']]+[c['decrypted']==c['message'] for c in d['other_chains']])")"
}
run host6_a "" 6 0
run dw8_a "" 8 64
run dw8_2cpus "0-1" 8 64
run single_a "" 1 0
run host6_b "" 6 0
run dw8_4cpus "0-3" 8 64
run single_b "" 1 0
