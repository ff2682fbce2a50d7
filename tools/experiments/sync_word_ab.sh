#!/bin/bash
# Waiting on a completion word the device writes (VPBS_SYNC_WORD=1, default) against waiting through the runtime (0), at the CPU shares a rank
# of an 8-GPU node may get.  Every run under its own timeout.   usage (GPU box): tools/experiments/sync_word_ab.sh OUTDIR [steps=200] [reps=2]
out=${1:-gpurun_out/sync_word_ab}; steps=${2:-200}; reps=${3:-2}
mkdir -p "$out"
run() {   # name cpus chains device_witness word
  local pre=""
  if [ "$2" != "16" ]; then pre="taskset -c 0-$(( $2 - 1 ))"; fi
  VPBS_SYNC_WORD=$5 VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=$3 VPBS_IVC_DEVICE_WITNESS=$4 timeout -k 5 300 $pre python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/$1.out" 2>&1
}
for rep in $(seq 1 $reps); do
  for b in 1 0; do
    run cpus2_dw_chains6_word${b}_$rep 2 6 64 $b
    run cpus4_dw_chains6_word${b}_$rep 4 6 64 $b
    run cpus16_host_chains6_word${b}_$rep 16 6 0 $b
    run cpus16_dw_chains6_word${b}_$rep 16 6 64 $b
    run cpus16_host_chains1_word${b}_$rep 16 1 0 $b
  done
done
python - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(os.path.join(out, "*.out"))):
    name = os.path.basename(f)[:-4]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        res[name] = "failed: %s" % e
        continue
    c = d.get("cpu_by_role") or {}
    res[name] = {"ms_per_chained_proof": round(d["ms_per_step"] / d["chains"], 3), "vpbs_per_s_equiv": round(d["chains"] * 1e3 / d["ms_per_step"] / 730, 4),
                 "cpu_ms_per_chained_proof": c.get("cpu_ms_per_chained_step"), "by_role": c.get("cpu_ms_per_chained_step_by_role"), "loadavg": round(d["host"]["loadavg"])}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in res.items():
    print(k, v if isinstance(v, str) else (v["ms_per_chained_proof"], v["vpbs_per_s_equiv"], v["cpu_ms_per_chained_proof"], v["by_role"].get("python") if v["by_role"] else None))
PY
