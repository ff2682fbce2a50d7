python -m pytest tests/test_gpu_gates.py -x -q -m gpu 2>&1 | tail -3; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/prof_gates2 && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gates2 -- python3 bench.py --workload step --steps 5 --warmup 1 --no-cpu-baseline --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc --no-survey-size --batch-chains 1 > gpurun_out/prof_gates.log 2>&1; python3 - <<EOP
import csv,glob,re
f=max(glob.glob("gpurun_out/prof_gates2/**/*_kernel_stats.csv",recursive=True))
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "gate_kernel" in n:
        print(re.sub(r"vpbs::\(anonymous namespace\)::","",n)[:40], r["Calls"], float(r["AverageNs"])/1e3); tot+=float(r["AverageNs"])/1e3
print("total", tot)
EOP
