cd "$GRAFT_REPO_ROOT"
F="--chains 1 --steps 80 --warmup 6 --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc"
for i in 1 2 3; do
  for pin in 1 0; do
    VPBS_POOL_PIN=$pin python3 bench.py $F 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['chain_ms_per_step_split']; print('pin=$pin', 'ms/step %.2f'%d['ms_per_step_proof'], 'late %.2f early %.2f prove %.2f'%(s['witness_late_phase_host'], s['witness_early_phase_on_a_second_thread'], s['prove_step']), 'load', d['host']['loadavg_1min'])"
  done
done
F4="--chains 4 --steps 60 --warmup 6 --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc"
for i in 1 2; do
  for pin in 1 0; do
    VPBS_POOL_PIN=$pin python3 bench.py $F4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['chain_ms_per_step_split']; print('4 chains pin=$pin', 'ms/proof %.2f'%d['ms_per_step_proof'], 'late %.2f early %.2f'%(s['witness_late_phase_host'], s['witness_early_phase_on_a_second_thread']), 'load', d['host']['loadavg_1min'])"
  done
done
