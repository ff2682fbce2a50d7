#!/bin/bash
# Run ON THE GPU BOX: bench.py's headline leg at 2 CPUs (taskset), device witness pipeline: the late phase on the host (default) against
# --device-late (the host generates no witness at all), three alternating runs each
F="--no-cpu-baseline --no-step-micro --no-single-chain --no-step-circuit --no-whole-pbs --no-survey-size --no-ivc --no-batch128"
for i in 1 2 3; do for late in "" "--device-late"; do
  taskset -c 0-$((${1:-2}-1)) python bench.py $F $late 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cpus=${1:-2} late=[$late] value', round(d['value'],4), 'ms_per_step', round(d['ms_per_step'],2), 'chains', d['config']['chains_per_gpu'])"
done; done
