#!/bin/bash
# Run ON THE GPU BOX: bench.py's headline leg under taskset CPUS (default 2) with 8 / 10 / 12 chains per GPU (device witness pipeline, late phase on the host)
F="--no-cpu-baseline --no-step-micro --no-single-chain --no-step-circuit --no-whole-pbs --no-survey-size --no-ivc --no-batch128"
CPUS=${1:-2}
for ch in 8 10 12; do
  taskset -c 0-$((CPUS-1)) python bench.py $F --chains $ch --detail /dev/null 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cpus=$CPUS chains=$ch value', d['value'], 'ms_per_step_proof', d['ms_per_step_proof'], 'witness', d['config']['witness'])"
done
