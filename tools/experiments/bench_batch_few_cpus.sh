#!/bin/bash
# bench.py's headline leg alone on 2 and 4 CPUs: the device pipeline's batch (steps whose early phases run on the device at once) 16 / 32 / 64
# and the default (half the timed steps: the window holds two whole batches)
F="--no-cpu-baseline --no-step-micro --no-single-chain --no-step-circuit --no-whole-pbs --no-survey-size --no-ivc --no-batch128"
for cpus in 2 4; do for b in -1 -1 ${1:-}; do
  s=$(date +%s)
  taskset -c 0-$((cpus-1)) python bench.py $F --device-witness $b 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cpus=$cpus batch=$b value', round(d['value'],4), 'ms_per_step', round(d['ms_per_step'],2), 'chains', d['config']['chains_per_gpu'], 'steps', d['steps'], 'warmup', d['warmup'], 'witness', d['config']['witness'])"
  echo "   wall $(( $(date +%s) - s )) s"
done; done
