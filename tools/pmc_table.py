#!/usr/bin/env python3
"""Condense the rocprofv3 counter CSVs written by tools/pmc_kernels.sh into one table per kernel (sums over launches).

Last column `valu_per_step_proof`: SQ_INSTS_VALU of ONE step proof, exact -- the dispatches between the first and the last
quotient_perm_kernel dispatch (that kernel runs once per step proof and never during setup) are whole periods of the step's kernel sequence,
so their sum / the number of periods is one step's count whatever ran before (the constants / sigmas commitment, the table kernels)."""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("vpbs::", "")
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"<.*", "", name) if "gate_kernel" not in name else name


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    per_step = collections.defaultdict(float)
    for d in sys.argv[1:]:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            seen = set()
            rows = list(csv.DictReader(open(path)))
            for r in rows:
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if d == sys.argv[1] and r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    launches[k] += 1
            marks = sorted({int(r["Dispatch_Id"]) for r in rows if short(r["Kernel_Name"]) == "quotient_perm_kernel"})
            if len(marks) >= 2 and any(r["Counter_Name"] == "SQ_INSTS_VALU" for r in rows) and not per_step:
                for r in rows:
                    if r["Counter_Name"] == "SQ_INSTS_VALU" and marks[0] < int(r["Dispatch_Id"]) <= marks[-1]:
                        per_step[short(r["Kernel_Name"])] += float(r["Counter_Value"]) / (len(marks) - 1)
    cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_SALU",
            "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"]
    print("kernel,launches," + ",".join(cols) + ",valu_issue_share,wait_inst_share,waitcnt_share,valu_per_step_proof")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        print(",".join(['"%s"' % k, str(launches[k])] + ["%.0f" % v.get(c, 0) for c in cols] +
                       ["%.3f" % (v.get("SQ_ACTIVE_INST_VALU", 0) / wc), "%.3f" % (v.get("SQ_WAIT_INST_ANY", 0) / wc), "%.3f" % (v.get("SQ_WAIT_ANY", 0) / wc),
                        "%.0f" % per_step.get(k, 0)]))


if __name__ == "__main__":
    main()
