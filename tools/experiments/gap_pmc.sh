cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gap_pmc; mkdir -p gpurun_out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/gap_pmc -- python3 tools/experiments/idle_gap.py > gpurun_out/gap_pmc.out 2> gpurun_out/gap_pmc.err
tail -c 300 gpurun_out/gap_pmc.err; cat gpurun_out/gap_pmc.out | cut -c1-100
python3 - <<'PY'
import csv, glob, collections
trace = glob.glob("gpurun_out/gap_pmc/**/*kernel_trace.csv", recursive=True)
cnt = glob.glob("gpurun_out/gap_pmc/**/*counter_collection.csv", recursive=True)
print(trace, cnt)
dur = {}
for r in csv.DictReader(open(trace[0])):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]))
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cnt[0])):
    vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
rows = []
for d, (name, ns, start) in dur.items():
    if "leaf_hash" in name and d in vals:
        rows.append((start, ns, vals[d].get("GRBM_GUI_ACTIVE", 0)))
rows.sort()
# print in groups of 30 to see the phases of the experiment
for i in range(0, len(rows), 30):
    grp = rows[i:i + 30]
    ns = sum(g[1] for g in grp) / len(grp); cyc = sum(g[2] for g in grp) / len(grp)
    print("leaf_hash launches %4d..: avg %.3f ms, GRBM_GUI_ACTIVE %.0f cycles -> %.0f MHz effective" % (i, ns / 1e6, cyc, cyc / ns * 1e3))
PY
