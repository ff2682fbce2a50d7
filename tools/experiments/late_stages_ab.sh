#!/bin/bash
# A-B-C of the late witness phase of a single IVC chain: one late phase after the proof (rounds 2-3) / staged by proof
# section with 8 strand threads / with 14.  Alternating repetitions, because the shared host's load moves on the scale of seconds.
# usage (on the GPU box): tools/experiments/late_stages_ab.sh OUTDIR [reps=5] [steps=200]
out=${1:-gpurun_out/late_ab}; reps=${2:-5}; steps=${3:-200}
mkdir -p "$out"
for rep in $(seq 1 "$reps"); do
  VPBS_IVC_LATE_STAGES=0 python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/unstaged_$rep.out" 2>&1
  VPBS_LATE_THREADS=8  python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/staged8_$rep.out" 2>&1
  VPBS_LATE_THREADS=14 python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/staged14_$rep.out" 2>&1
done
python - "$out" <<'PY'
import glob, json, os, statistics, sys
out = sys.argv[1]
res = {}
for cfg in ("unstaged", "staged8", "staged14"):
    rows = []
    for f in sorted(glob.glob(os.path.join(out, cfg + "_*.out"))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d["ms_per_step_split"]
        rows.append((d["ms_per_step"], s["witness_late_phase_host"], s["late_rows_to_device"], s["prove_step"],
                     s.get("late_stages_run_during_the_previous_proofs_fri_stage", 0.0), d["host"]["loadavg"]))
    med = lambda i: round(statistics.median(r[i] for r in rows), 3)
    res[cfg] = {"runs": len(rows), "ms_per_step_median": med(0), "ms_per_step_min": round(min(r[0] for r in rows), 3), "late_on_critical_path_median": med(1),
                "scatter_median": med(2), "prove_step_median": med(3), "late_ahead_median": med(4), "loadavg_median": med(5),
                "ms_per_step_runs": [round(r[0], 2) for r in rows]}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
