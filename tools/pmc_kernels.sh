#!/bin/bash
# Run ON THE GPU BOX: SQ counters of every kernel of a short bench run, one rocprofv3 --pmc pass per counter group (kernel trace only).
# usage: bash tools/pmc_kernels.sh   -> gpurun_out/pmc_a, pmc_b (csv), pmc_shared (the first group again with the settings of a context that
# SHARES the GPU with other chains: 16-lane Poseidon threshold 2048, PoW in rounds -- what the eight-chain headline runs); condense with tools/pmc_table.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_shared
CMD="python3 bench.py --workload step --steps 3 --warmup 1 --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc --batch-chains 1 --detail /dev/null"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_a -- $CMD > /dev/null 2> gpurun_out/pmc_a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_b -- $CMD > /dev/null 2> gpurun_out/pmc_b.err
VPBS_WIDE_THRESHOLD=2048 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_shared -- $CMD > /dev/null 2> gpurun_out/pmc_shared.err
tail -c 300 gpurun_out/pmc_a.err; tail -c 300 gpurun_out/pmc_b.err
ls gpurun_out/pmc_a/*/ gpurun_out/pmc_b/*/ 2>/dev/null | head
