cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
F="--no-cpu-baseline --no-step-micro --no-single-chain --no-step-circuit --no-whole-pbs --no-survey-size --no-ivc --no-batch128"
rm -rf gpurun_out/prof_walk
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_walk -- python3 bench.py $F --device-witness 30 --device-late --chains 1 --steps 30 --warmup 4 --detail /dev/null > /dev/null 2> gpurun_out/prof_walk.err
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_walk/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'wd_walk_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(len(rows),'walk launches')
# group by grid size / order: print a window of 18 launches in the middle
mid=len(rows)//2
mid-=mid%6
for r in rows[mid:mid+18]:
    print(r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or '', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us  start+', (int(r['Start_Timestamp'])-int(rows[mid]['Start_Timestamp']))/1e3)
PY
