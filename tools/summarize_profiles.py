#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/ into the committed summaries under profiles/.

usage: tools/summarize_profiles.py <round-tag> <stats_dir> <fetch_dir> <write_dir>
  stats_dir : rocprofv3 --kernel-trace --stats --output-format csv  -- python3 bench.py ...
  fetch_dir : rocprofv3 --pmc FETCH_SIZE --kernel-trace ...         (own pass)
  write_dir : rocprofv3 --pmc WRITE_SIZE --kernel-trace ...         (own pass)
HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = "leaf_hash_kernel"


GATE_KINDS = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext",
              "reducing", "reducing_ext", "random_access", "exponentiation", "coset_interpolation"]


def short(name):
    import re
    g = re.search(r"gate_kernel<(\d+)u", name)
    if g:
        return "gate_kernel<%s>" % GATE_KINDS[int(g.group(1))]
    m = re.search(r"(\w+)\s*(<[^(]*>)?\(", name.replace("(anonymous namespace)", "anon"))
    return m.group(1) if m else name


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    assert files, pattern
    return max(files, key=os.path.getmtime)   # gpurun merges runs into one directory: take the newest


def counter_rows(d, counter):
    rows = list(csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))))
    rows = [r for r in rows if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    dominant = sys.argv[5] if len(sys.argv) > 5 else DOMINANT
    out_dir = os.path.join(ROOT, "profiles")
    # 1. kernel stats table
    rows = list(csv.DictReader(open(one(os.path.join(stats_dir, "**", "*_kernel_stats.csv")))))
    with open(os.path.join(out_dir, "%s_rocprof_kernel_stats.csv" % tag), "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
        for r in rows:
            f.write("%s,%s,%.3f,%.1f,%s,%.1f,%.1f\n" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                        float(r["AverageNs"]) / 1e3, r["Percentage"], int(r["MinNs"]) / 1e3,
                                                        int(r["MaxNs"]) / 1e3))
    # per-launch durations of the dominant kernel from the trace of the same run, in launch order
    trace = list(csv.DictReader(open(one(os.path.join(stats_dir, "**", "*_kernel_trace.csv")))))
    lh = [r for r in trace if short(r["Kernel_Name"]) == dominant]
    lh.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in lh]
    # launch order: constants_sigmas (85 cols) once, then wires(135) / zs_pp(20) / quotient(16) per step proof
    kinds = ["constants_sigmas"] + ["wires", "zs_partial_products", "quotient"] * ((len(durs) - 1) // 3)
    by_kind = collections.defaultdict(list)
    for k, d in zip(kinds, durs):
        by_kind[k].append(d)
    # 2. PMC traffic of the dominant kernel
    fetch = [float(r["Counter_Value"]) for r in counter_rows(fetch_dir, "FETCH_SIZE") if short(r["Kernel_Name"]) == dominant]
    write = [float(r["Counter_Value"]) for r in counter_rows(write_dir, "WRITE_SIZE") if short(r["Kernel_Name"]) == dominant]
    kinds_p = ["constants_sigmas"] + ["wires", "zs_partial_products", "quotient"] * ((len(fetch) - 1) // 3)
    cols = {"constants_sigmas": 86, "wires": 135, "zs_partial_products": 20, "quotient": 16}
    lde = 1 << 19   # bench.py default: degree 2^16, LDE 2^19
    per_kind = {}
    for kind in cols:
        fv = [v for k, v in zip(kinds_p, fetch) if k == kind]
        wv = [v for k, v in zip(kinds_p, write) if k == kind]
        if not fv:
            continue
        hbm = (2 * sum(fv) / len(fv) + sum(wv) / len(wv)) * 1024
        alg = lde * (cols[kind] * 8 + 32)
        per_kind[kind] = {"launches": len(fv), "FETCH_SIZE_KiB_raw_avg": sum(fv) / len(fv), "WRITE_SIZE_KiB_avg": sum(wv) / len(wv),
                          "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg,
                          "avg_duration_us": sum(by_kind[kind]) / max(1, len(by_kind[kind]))}
    step_kinds = ("wires", "zs_partial_products", "quotient")
    summary = {
        "kernel": dominant,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), gfx950 correction: FETCH_SIZE x2",
        "per_launch_kind": per_kind,
        "hbm_bytes_per_launch_avg": sum(per_kind[k]["hbm_bytes_per_launch"] for k in step_kinds) / 3,
        "algorithmic_bytes_per_launch_avg": sum(per_kind[k]["algorithmic_bytes_per_launch"] for k in step_kinds) / 3,
        "avg_duration_us_per_step_launch": sum(per_kind[k]["avg_duration_us"] for k in step_kinds) / 3,
    }
    with open(os.path.join(out_dir, "%s_pmc_leaf_hash.json" % tag), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))
    # 3. the gate-constraint stage: HBM traffic of the one-launch kernel (or of the per-gate launches) against its algorithmic bytes
    def traffic(pred):
        fv = [float(r["Counter_Value"]) for r in counter_rows(fetch_dir, "FETCH_SIZE") if pred(short(r["Kernel_Name"]))]
        wv = [float(r["Counter_Value"]) for r in counter_rows(write_dir, "WRITE_SIZE") if pred(short(r["Kernel_Name"]))]
        tr = [r for r in trace if pred(short(r["Kernel_Name"]))]
        return fv, wv, [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
    fused = traffic(lambda k: k == "gate_fused_kernel" or k == "gate_tile_kernel")
    per_gate = traffic(lambda k: k.startswith("gate_kernel<"))
    steps = len(by_kind["wires"])
    gates = {"algorithmic_bytes_per_step": lde * (135 + 6) * 8,
             "note": "algorithmic = every column a gate can read, once: 135 wire + 6 selector / gate-constant columns x 2^19 x 8 B (the sigma and Z "
                     "columns belong to the permutation part, quotient_perm_kernel; the output planes -- one plane of 2 x 2^19 x 8 B for "
                     "gate_tile_kernel, n_items planes for gate_fused_kernel -- and the alpha powers are not counted); measured = FETCH_SIZE x 2 + WRITE_SIZE "
                     "per step (FETCH_SIZE counts what leaves the XCD's L2, whether the memory-side cache or HBM serves it)"}
    one_launch_name = "gate_tile_kernel" if any(short(r["Kernel_Name"]) == "gate_tile_kernel" for r in trace) else "gate_fused_kernel"
    for name, (fv, wv, du) in ((one_launch_name, fused), ("per_gate_kernels", per_gate)):
        if fv:
            hbm = (2 * sum(fv) + sum(wv)) * 1024 / max(1, steps)
            gates[name] = {"launches": len(fv), "hbm_bytes_per_step": hbm, "ratio_to_algorithmic": hbm / gates["algorithmic_bytes_per_step"],
                           "fetch_bytes_per_step": 2 * sum(fv) * 1024 / max(1, steps), "write_bytes_per_step": sum(wv) * 1024 / max(1, steps),
                           "duration_us_per_step": sum(du) / max(1, steps)}
    with open(os.path.join(out_dir, "%s_pmc_gates.json" % tag), "w") as f:
        json.dump(gates, f, indent=1)
    print(json.dumps(gates, indent=1))


if __name__ == "__main__":
    main()
