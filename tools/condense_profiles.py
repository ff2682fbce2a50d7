#!/usr/bin/env python3
"""After `gpurun -- 'bash tools/refresh_profiles.sh'` has merged its output into gpurun_out/: condense it into profiles/<tag>_* (run HERE).
gpurun merges, it does not replace: every rocprofv3 output directory may hold the files of earlier calls next to the newest process's, so the
older ones are dropped first.  usage: python tools/condense_profiles.py r06 [--bench gpurun_out/bench_default.json]"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def newest_process_only(name):
    base = os.path.join(OUT, name, "runc")
    infos = glob.glob(base + "/*_agent_info.csv")
    if not infos:
        return False
    pid = os.path.basename(max(infos, key=os.path.getmtime)).split("_")[0]
    for f in glob.glob(base + "/*"):
        if not os.path.basename(f).startswith(pid + "_"):
            os.remove(f)
    return True


def table(dirs, dest):
    dirs = [os.path.join(OUT, d) for d in dirs if os.path.isdir(os.path.join(OUT, d))]
    if dirs:
        with open(os.path.join(PROF, dest), "w") as f:
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_table.py")] + dirs, stdout=f, check=True)


def main():
    tag = sys.argv[1]
    bench = sys.argv[sys.argv.index("--bench") + 1] if "--bench" in sys.argv else os.path.join(OUT, "bench_default.json")
    for d in ("pmc_a", "pmc_b", "pmc_shared", "pmc2048_a", "prof_fetch", "prof_write", "prof_stats", "prof_n2048", "prof_chain1", "prof_chain8"):
        newest_process_only(d)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), tag] +
                   [os.path.join(OUT, d) for d in ("prof_stats", "prof_fetch", "prof_write")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    table(["pmc_a", "pmc_b"], tag + "_pmc_sq_kernels.csv")
    table(["pmc_shared"], tag + "_pmc_sq_kernels_shared_gpu.csv")
    table(["pmc2048_a"], tag + "_pmc_sq_kernels_n2048.csv")
    copies = {"prof_chain1_kernel_stats.csv": "_rocprof_chain1_kernel_stats.csv", "prof_chain8_kernel_stats.csv": "_rocprof_chain8_kernel_stats.csv",
              "prof_n2048_kernel_stats.csv": "_rocprof_ivc_chain_n2048_kernel_stats.csv", "prof_chain1_bench.json": "_bench_chain1_under_rocprof.json",
              "prof_chain8_bench.json": "_bench_chain8_under_rocprof.json", "prof_stats_bench.json": "_bench_step_under_rocprof.json",
              "sharded_replay.json": "_sharded_rank_times.json", "ivc_chain_n2048_full.json": "_ivc_chain_n2048.json",
              "gate_times.json": "_gate_times.json", "pmc_sources.json": "_pmc_sources.json", "bench_default.line": "_bench_line.json"}
    for src, dst in copies.items():
        if os.path.exists(os.path.join(OUT, src)):
            shutil.copyfile(os.path.join(OUT, src), os.path.join(PROF, tag + dst))
    if os.path.exists(bench):
        shutil.copyfile(bench, os.path.join(PROF, tag + "_bench_detail_n1.json"))
    print("condensed into profiles/%s_*; now check the figures quoted in DESIGN.md / README.md / profiles/README.md" % tag)


if __name__ == "__main__":
    main()
