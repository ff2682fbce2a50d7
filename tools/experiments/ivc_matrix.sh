#!/bin/bash
# Run ON THE GPU BOX: whole-chain throughput of tools/prove_ivc.py over a MATRIX of host configurations -- the one runner behind the round-4/5
# host-side findings (CPU share of a rank, chains per GPU, hardware queues, sleeping waits, device witness batch).  One line per run:
# ms per chained proof, vPBS/s, late phase / proof split, CPU-ms per proof (VPBS_CPU_BY_ROLE), host load.
# usage: [BY_ROLE=1] ivc_matrix.sh RUN [RUN ...] | ivc_matrix.sh --preset NAME     (BY_ROLE=1 adds the CPU-ms per proof of every thread role)
#   RUN = "[VAR=value,VAR=value:]cpus:chains:device_witness:steps"   (cpus = taskset mask size, 16 = no mask)
# Presets = the argument lists of the one-off scripts this replaces (their names are what profiles/README.md and older records cite):
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
declare -A PRESET=(
  [cpu_share_now]="2:6:64:250 2:6:64:250 2:8:64:250 4:6:64:250 4:8:64:250 4:8:64:250 16:8:64:200 16:6:0:200"
  [few_cpus_repeat]="2:6:64:300 2:6:64:200 2:6:64:300 2:6:64:200 4:6:64:300 4:6:64:200 16:6:64:300 2:6:64:730"
  [hw_queues_dw]="GPU_MAX_HW_QUEUES=8:16:6:64:200 GPU_MAX_HW_QUEUES=8:4:6:64:200 GPU_MAX_HW_QUEUES=16:16:6:64:200 GPU_MAX_HW_QUEUES=16:4:6:64:200 GPU_MAX_HW_QUEUES=24:16:6:64:200 GPU_MAX_HW_QUEUES=24:4:6:64:200 GPU_MAX_HW_QUEUES=16:16:6:0:200 GPU_MAX_HW_QUEUES=8:16:6:0:200"
  [hw_queues_dw2]="GPU_MAX_HW_QUEUES=16:4:8:64:200 GPU_MAX_HW_QUEUES=16:4:10:64:200 GPU_MAX_HW_QUEUES=16:2:6:64:200 GPU_MAX_HW_QUEUES=16:2:8:64:200 GPU_MAX_HW_QUEUES=12:4:6:64:200 GPU_MAX_HW_QUEUES=16:16:8:64:200 GPU_MAX_HW_QUEUES=16:4:6:32:200"
  [more_chains_few_cpus]="2:10:64:200 2:12:64:200 2:8:64:200 4:10:64:200 4:12:64:200 2:10:32:200"
  [host8_queues]="GPU_MAX_HW_QUEUES=8:16:8:0:200 GPU_MAX_HW_QUEUES=12:16:8:0:200 GPU_MAX_HW_QUEUES=16:16:8:0:200 VPBS_LATE_THREADS=6:16:8:0:200"
  [host_chains_16cpus]="16:8:0:200 16:9:0:200 16:10:0:200 16:12:0:200"
  [auto_blocking_ab]="16:6:0:200 VPBS_BLOCKING_SYNC=0:16:6:0:200 16:1:0:300 VPBS_BLOCKING_SYNC=0:16:1:0:300 8:8:64:200"
  [blocking_at_8_cpus]="VPBS_BLOCKING_SYNC=0:8:8:64:200 VPBS_BLOCKING_SYNC=1:8:8:64:200 VPBS_BLOCKING_SYNC=0:10:8:64:200 VPBS_BLOCKING_SYNC=1:10:8:64:200"
)
runs=("$@")
if [ "$1" = "--preset" ]; then read -r -a runs <<< "${PRESET[$2]}"; [ ${#runs[@]} -gt 0 ] || { echo "presets: ${!PRESET[*]}"; exit 2; }; fi
for run in "${runs[@]}"; do
  IFS=: read -r -a f <<< "$run"
  envs=""; [ ${#f[@]} -eq 5 ] && { envs="${f[0]//,/ }"; f=("${f[@]:1}"); }
  cpus=${f[0]} chains=${f[1]} dw=${f[2]} steps=${f[3]}
  pre=""; [ "$cpus" != "16" ] && pre="taskset -c 0-$(( cpus - 1 ))"
  env $envs VPBS_CPU_BY_ROLE=1 VPBS_IVC_CHAINS=$chains VPBS_IVC_DEVICE_WITNESS=$dw timeout -k 5 400 $pre python3 tools/prove_ivc.py 1024 728 16 $steps 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; c=d.get('cpu_by_role') or {}
print('[$envs] cpus=$cpus chains=$chains dw=$dw', round(d['ms_per_step']/d['chains'],3), 'ms/proof', round(d['chains']*1e3/d['ms_per_step']/730,4), 'vPBS/s  late', round(s['witness_late_phase_host'],2), 'prove', round(s['prove_step'],2), 'cpu-ms/proof', c.get('cpu_ms_per_chained_step'), 'load', round(d['host']['loadavg']))
if '${BY_ROLE:-}': print('    by role', c.get('cpu_ms_per_chained_step_by_role'))"
done
