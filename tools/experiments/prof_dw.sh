# Run ON THE GPU BOX: kernel statistics of the device witness pipeline (one chain: every kernel alone)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_dw
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/prof_dw -- python3 bench.py --chains ${1:-1} --device-witness ${2:-64} --steps 128 --warmup 6 \
  --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc \
  > gpurun_out/prof_dw_bench.json 2> gpurun_out/prof_dw.err
f=$(ls gpurun_out/prof_dw/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_dw_kernel_stats.csv && head -40 "$f" | cut -c1-200
python3 -c "
import json; d=json.load(open('gpurun_out/prof_dw_bench.json')); print('value', d['value'], 'ms_per_step_proof', d['ms_per_step_proof'], d['chain_ms_per_step_split'])"
