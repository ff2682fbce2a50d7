#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles_r03.sh'): every rocprofv3 pass behind profiles/r03_*.  Counter passes are their
# own runs (kernel trace only).  Condensed afterwards in the authoring container (profiles/ is tracked):
#   python tools/summarize_profiles.py r03 gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
#   python tools/pmc_table.py gpurun_out/pmc_a gpurun_out/pmc_b > profiles/r03_pmc_sq_kernels.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
# the synthetic step (one kernel at a time on the device): kernel statistics, then HBM traffic of the leaf-hash and gate kernels
CMD="python3 bench.py --workload step --steps 10 --warmup 2 --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc --batch-chains 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $CMD > gpurun_out/prof_stats_bench.json 2> gpurun_out/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- $CMD > /dev/null 2> gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- $CMD > /dev/null 2> gpurun_out/prof_write.err
bash tools/pmc_kernels.sh > /dev/null 2>&1
# the headline workload (chained step proofs of the cyclic circuit): one chain (every kernel alone), then the headline's four chains
bash tools/prof_chain.sh 1 chain1 | tail -2
bash tools/prof_chain.sh 6 chain6 | tail -2
