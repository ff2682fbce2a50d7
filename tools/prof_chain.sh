#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/prof_chain.sh [chains] [tag]'): rocprofv3 kernel-trace statistics of the headline workload -- chained
# step proofs of the cyclic circuit through vpbs_ivc_prove_pbs (bench.py's default workload with the secondary legs switched off).
# One chain: every kernel alone on the device (the per-kernel averages the roofline object is priced with); three chains: the headline mix.
CH=${1:-1}
TAG=${2:-chain${CH}}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8   # what bench.py sets for itself; under the profiler the runtime is loaded before bench.py starts
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --chains $CH --steps 60 --warmup 6 \
  --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc \
  --detail gpurun_out/prof_${TAG}_bench.json > /dev/null 2> gpurun_out/prof_${TAG}.err
f=$(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_${TAG}_kernel_stats.csv && head -25 "$f"
python3 -c "
import json; d=json.load(open('gpurun_out/prof_${TAG}_bench.json')); print('value', d['value'], 'ms_per_step_proof', d['ms_per_step_proof'], 'leaf_hash ms/step', d['roofline']['kernel_ms_per_step'], 'frac', d['roofline']['frac'])"
