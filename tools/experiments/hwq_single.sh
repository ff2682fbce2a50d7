cd "$GRAFT_REPO_ROOT"
F="--steps 80 --warmup 6 --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc --chains 1 --device-witness 0"
for i in 1 2 3; do
for q in 4 8; do
    GPU_MAX_HW_QUEUES=$q python3 bench.py $F 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['chain_ms_per_step_split']; print('hwq=$q single chain', 'ms/step %.2f'%d['ms_per_step_proof'], 'late %.2f early %.2f prove %.2f'%(s['witness_late_phase_host'], s['witness_early_phase_on_a_second_thread'], s['prove_step']), 'load %.0f'%d['host']['loadavg_1min'])"
done
done
