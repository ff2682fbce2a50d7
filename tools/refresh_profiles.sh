#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh'): the rocprofv3 passes behind profiles/r01_*.
# Each counter pass is its own run (kernel trace only); outputs land in gpurun_out/ and are condensed by
# tools/summarize_profiles.py afterwards (run that in the authoring container: profiles/ is tracked).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
CMD="python3 bench.py --workload step --steps 10 --warmup 2 --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc --batch-chains 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $CMD > gpurun_out/prof_stats_bench.json 2> gpurun_out/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- $CMD > /dev/null 2> gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- $CMD > /dev/null 2> gpurun_out/prof_write.err
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 300 gpurun_out/bench_default.err
python3 -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['batch'], d['cpu_baseline']['ms_per_step'])"
