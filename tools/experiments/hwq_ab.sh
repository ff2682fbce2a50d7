# Run ON THE GPU BOX: GPU_MAX_HW_QUEUES (HIP runtime: hardware queues the process's streams are mapped onto) against the multi-chain legs
cd "$GRAFT_REPO_ROOT"
F="--steps 60 --warmup 6 --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc"
for i in 1 2; do
for q in default 8 16 24; do
  for dw in 0 64; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python3 bench.py $F --chains 6 --device-witness $dw 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['chain_ms_per_step_split']; print('hwq=$q dw=$dw', 'value %.4f'%d['value'], 'ms/proof %.2f'%d['ms_per_step_proof'], 'late %.2f scatter %.2f prove %.2f'%(s['witness_late_phase_host'], s['late_rows_to_device'], s['prove_step']), 'load %.0f'%d['host']['loadavg_1min'])"
  done
done
done
