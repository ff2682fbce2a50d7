# Run ON THE GPU BOX: the headline as a function of the CPUs a rank may use (what one rank of an 8-GPU node gets when the host is small),
# with the early witness phases on the host and on the device.  The affinity mask stands in for the share; CHAINS / DW override.
cd "$GRAFT_REPO_ROOT"
F="--steps 60 --warmup 6 --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc"
for cpus in ${CPUS:-2 4 8 16}; do
  for dw in ${DW:-0 32}; do
    for ch in ${CHAINS:-0}; do
      taskset -c 0-$((cpus-1)) python3 bench.py $F --device-witness $dw --chains $ch 2>gpurun_out/cpu_share.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['chain_ms_per_step_split']; print('cpus=$cpus dw=$dw chains', d['config']['chains_per_gpu'], 'value %.4f'%d['value'], 'ms/proof %.2f'%d['ms_per_step_proof'], 'late %.2f early %.2f prove %.2f'%(s['witness_late_phase_host'], s['witness_early_phase_on_a_second_thread'], s['prove_step']), 'load %.0f'%d['host']['loadavg_1min'])" || tail -3 gpurun_out/cpu_share.err
    done
  done
done
