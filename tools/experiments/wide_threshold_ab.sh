#!/bin/bash
# the 16-lane Poseidon form for launches of few independent permutations: where is the break-even at degree 2^16 (FRI round-0 tree: 2^15 leaves of
# four dependent permutations; Merkle levels of 2^15 / 2^14 parents)?  synthetic step, HIP-event kernel groups + wall per step proof
run() { # fri_thr wide_thr
  VPBS_FRI_LEAF_WIDE_THRESHOLD=$1 VPBS_WIDE_THRESHOLD_ONLY=$2 python bench.py --workload step --steps 30 --warmup 5 --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc --batch-chains 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_one_step') or d.get('step_micro',{}).get('kernel_ms_one_step') or {}; ms=d.get('ms_per_step_proof') or d.get('step_micro',{}).get('ms_per_step_proof') or d.get('ms_per_step'); print('fri_thr=$1 wide_thr=$2: %.3f ms/step proof; fri_tree %.4f merkle_levels %.4f pow %.4f' % (ms, k.get('fri_tree',-1), k.get('merkle_levels',-1), k.get('pow_search',-1)))"
}
for rep in 1 2; do
  run 16384 16384; run 32768 16384; run 65536 16384
done
