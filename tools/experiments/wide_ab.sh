# Run ON THE GPU BOX: the 16-lane / one-lane Poseidon crossover (VPBS_OPT_WIDE_THRESHOLD) with six chains per GPU, where latency is hidden
# by the other chains and only the instruction count of a form matters (the 16-lane form issues 3.7 x the instructions per permutation)
cd "$GRAFT_REPO_ROOT"
F="--steps 60 --warmup 6 --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc --device-witness 0"
for i in 1 2; do
for th in ${TH:-16384 4096 1024 256}; do
  for ch in ${CHAINS:-6}; do
    VPBS_WIDE_THRESHOLD=$th python3 bench.py $F --chains $ch 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide_threshold=$th chains=$ch', 'value %.4f'%d['value'], 'ms/proof %.2f'%d['ms_per_step_proof'], 'load %.0f'%d['host']['loadavg_1min'])"
  done
done
done
