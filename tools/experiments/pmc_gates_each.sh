#!/bin/bash
# Run ON THE GPU BOX: wave-level VALU instructions of gate_tile_kernel for every gate type ALONE (tools/time_gates.py launches them in a fixed
# order: each single gate twice, then the 14 standard gates, then the cyclic circuit's 13) -- which evaluators carry the 0.43 G of the step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_gates_each
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_gates_each -- python3 tools/time_gates.py 16 1 > gpurun_out/pmc_gates_each.json 2> gpurun_out/pmc_gates_each.err
python3 - <<'P'
import csv, glob
f = glob.glob('gpurun_out/pmc_gates_each/*/*counter_collection.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'gate_tile_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'SQ_INSTS_VALU']
names = ["constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing", "reducing_ext",
         "random_access", "exponentiation", "coset_interpolation", "all_14", "cyclic_13", "all_14_3lanes", "cyclic_13_3lanes"]
by = {}
for r in rows: by.setdefault(r['Dispatch_Id'], 0.0); by[r['Dispatch_Id']] += float(r['Counter_Value'])
vals = [by[k] for k in sorted(by, key=int)]
print(len(vals), 'gate_tile dispatches')
for i, n in enumerate(names):
    if 2 * i + 1 < len(vals): print('%-22s %.4f G wave-instructions, %.0f per wave (8 waves per 64-point tile)' % (n, vals[2 * i + 1] / 1e9, vals[2 * i + 1] / 65536))
P
