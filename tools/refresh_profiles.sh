#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh [quick]'): the rocprofv3 passes behind profiles/rNN_*.  Counter passes are
# their own runs (kernel trace only).  Condensed afterwards in the authoring container by `python tools/condense_profiles.py rNN` (drops what
# earlier calls left in gpurun_out/, then summarize_profiles.py + pmc_table.py for the three counter tables + the copies into profiles/).
# Every bench run here writes its full result with --detail (stdout carries only the compact record).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
python3 tools/kernel_sources.py > gpurun_out/pmc_sources.json          # the device code these counters belong to
CMD="python3 bench.py --workload step --steps 10 --warmup 2 --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc --batch-chains 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $CMD --detail gpurun_out/prof_stats_bench.json > /dev/null 2> gpurun_out/prof_stats.err
f=$(ls gpurun_out/prof_stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/prof_step_kernel_stats.csv
bash tools/pmc_kernels.sh > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- $CMD --detail /dev/null > /dev/null 2> gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- $CMD --detail /dev/null > /dev/null 2> gpurun_out/prof_write.err
[ "$1" = "quick" ] && exit 0
bash tools/prof_chain.sh 1 chain1 | tail -2
bash tools/prof_chain.sh 8 chain8 | tail -2
export GPU_MAX_HW_QUEUES=8
rm -rf gpurun_out/prof_n2048 gpurun_out/pmc2048_a
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n2048 -- python3 tools/prove_ivc.py 2048 728 17 24 > gpurun_out/prof_n2048_chain.json 2> gpurun_out/prof_n2048.err
f=$(ls gpurun_out/prof_n2048/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/prof_n2048_kernel_stats.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc2048_a -- python3 tools/prove_ivc.py 2048 728 17 6 > /dev/null 2> gpurun_out/pmc2048_a.err
python3 tools/prove_ivc.py 2048 728 17 730 > gpurun_out/ivc_chain_n2048_full.json 2> gpurun_out/ivc_chain_n2048_full.err
python3 bench.py --mode sharded-replay --steps 10 > gpurun_out/sharded_replay.json 2> gpurun_out/sharded_replay.err
python3 tools/time_gates.py 16 10 > gpurun_out/gate_times.json 2> /dev/null
python3 bench.py --detail gpurun_out/bench_default.json > gpurun_out/bench_default.line 2> gpurun_out/bench_default.err
tail -c 400 gpurun_out/bench_default.err; cat gpurun_out/bench_default.line
