#!/bin/bash
# Six chains per GPU (the headline arrangement): does the late-phase arrangement of the single chain cost throughput?  A: round 3 (one late
# phase, 8 threads); B: stages + strands, 8 threads; C: stages + strands, 14 threads; D: stages + strands, 4 threads.  Alternating repetitions.
# usage (GPU box): tools/experiments/chains6_late_ab.sh OUTDIR [reps=3] [steps=240]
out=${1:-gpurun_out/chains6_ab}; reps=${2:-3}; steps=${3:-240}
mkdir -p "$out"
for rep in $(seq 1 "$reps"); do
  VPBS_IVC_CHAINS=6 VPBS_IVC_LATE_STAGES=0 VPBS_LATE_THREADS=8 python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/A_unstaged8_$rep.out" 2>&1
  VPBS_IVC_CHAINS=6 VPBS_LATE_THREADS=8  python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/B_staged8_$rep.out" 2>&1
  VPBS_IVC_CHAINS=6 VPBS_LATE_THREADS=14 python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/C_staged14_$rep.out" 2>&1
  VPBS_IVC_CHAINS=6 VPBS_LATE_THREADS=4  python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/D_staged4_$rep.out" 2>&1
done
python - "$out" <<'PY'
import glob, json, os, statistics, sys
out = sys.argv[1]
res = {}
for cfg in ("A_unstaged8", "B_staged8", "C_staged14", "D_staged4"):
    rows = []
    for f in sorted(glob.glob(os.path.join(out, cfg + "_*.out"))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows.append((d["ms_per_step"] / 6, d["host"]["loadavg"]))
    res[cfg] = {"ms_per_chained_proof_runs": [round(r[0], 3) for r in rows], "median": round(statistics.median(r[0] for r in rows), 3),
                "loadavg": [round(r[1]) for r in rows]}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
