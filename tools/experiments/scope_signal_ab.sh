#!/bin/bash
# ROC_SYSTEM_SCOPE_SIGNAL=0 (the HIP runtime creates its completion signals without an interrupt event: the HSA async-events thread, which
# otherwise spends 0.5-0.8 CPU in KFD event waits while the device is busy, has nothing to wait on) against the default, at the CPU shares a rank
# of an 8-GPU node may get.   usage (GPU box): tools/experiments/scope_signal_ab.sh OUTDIR [steps=200]
out=${1:-gpurun_out/scope_ab}; steps=${2:-200}
mkdir -p "$out"
run() {   # name cpus chains device_witness scope
  local pre=""
  if [ "$2" != "16" ]; then pre="taskset -c 0-$(( $2 - 1 ))"; fi
  ROC_SYSTEM_SCOPE_SIGNAL=$5 VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=$3 VPBS_IVC_DEVICE_WITNESS=$4 $pre python tools/prove_ivc.py 1024 728 16 "$steps" > "$out/$1.out" 2>&1
}
for rep in 1 2; do
  for b in 1 0; do
    run cpus2_dw_chains6_scope${b}_$rep 2 6 64 $b
    run cpus4_dw_chains6_scope${b}_$rep 4 6 64 $b
    run cpus16_host_chains6_scope${b}_$rep 16 6 0 $b
    run cpus16_dw_chains6_scope${b}_$rep 16 6 64 $b
    run cpus16_host_chains1_scope${b}_$rep 16 1 0 $b
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
                 "verified": d.get("verified"), "decrypted": d.get("decrypted"),
                 "cpu_ms_per_chained_proof": c.get("cpu_ms_per_chained_step"), "by_role": c.get("cpu_ms_per_chained_step_by_role"), "loadavg": round(d["host"]["loadavg"])}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in res.items():
    print(k, v if isinstance(v, str) else (v["ms_per_chained_proof"], v["vpbs_per_s_equiv"], v["cpu_ms_per_chained_proof"], v["verified"]))
PY
