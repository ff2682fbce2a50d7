F="--no-cpu-baseline --no-step-micro --no-single-chain --no-step-circuit --no-whole-pbs --no-survey-size --no-ivc --no-batch128"
for late in "" "--device-late"; do
  VPBS_TRACE_IVC=1 VPBS_CPU_BY_ROLE=1 taskset -c 0-1 python bench.py $F $late --detail gpurun_out/r6_fewcpu2${late}.json > gpurun_out/r6_fewcpu2${late}.line 2> gpurun_out/r6_fewcpu2${late}.err
  cut -c1-300 gpurun_out/r6_fewcpu2${late}.line; grep "ivc device witness\|cpu by role\|CPU" gpurun_out/r6_fewcpu2${late}.err | head -12
done
