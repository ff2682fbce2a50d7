# Run ON THE GPU BOX: the headline at 3 / 4 / 5 / 6 chains per GPU (same box, interleaved)
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for ch in ${CHAINS:-4 5 6 3}; do
    python3 bench.py --chains $ch --steps 60 --warmup 6 --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['chain_ms_per_step_split']; print('chains=$ch', 'value %.4f'%d['value'], 'ms/proof %.2f'%d['ms_per_step_proof'], 'late %.2f early %.2f'%(s['witness_late_phase_host'], s['witness_early_phase_on_a_second_thread']), 'load %.0f'%d['host']['loadavg_1min'])"
  done
done
