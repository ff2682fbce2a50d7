"""The bench LINE and the bench DETAIL.

`python bench.py` measures a dozen legs and used to print all of them, with their explanations, on one JSON line: 21 kB in round 5, and that
round the driver stored `parsed: null`.  The line is a RECORD: numbers and short identifiers, at most LINE_LIMIT bytes.  Everything else -- the
legs, per-kernel tables, the sentences that explain them -- is the DETAIL, written to bench_detail.json next to bench.py (and to
gpurun_out/bench_detail.json where that directory exists, so it travels back from the GPU box).

No torch, no prover library: the CPU suite builds lines from recorded details with this module alone.
"""
import json
import os

LINE_LIMIT = 4096
METRIC = "vPBS proofs/sec at N=1024 (1/2/4/8 GPU); prover ms/proof vs CPU baseline"      # BASELINE.json "metric", verbatim
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _num(x, digits=6):
    """a number as the line carries it: finite, six significant digits (None stays None)"""
    if x is None or isinstance(x, (bool, str)):
        return x
    if isinstance(x, int):
        return x
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _short(s, n):
    s = " ".join(str(s).split())
    return s if len(s) <= n else s[:n - 1].rstrip() + "~"


def workload_id(d):
    """config.workload in <= 200 characters: what a step is, on what, through which entry point"""
    cfg = d.get("config", {})
    if cfg.get("workload_id"):
        return _short(cfg["workload_id"], 200)
    w = str(cfg.get("workload", ""))
    if "vpbs_ivc_prove_pbs" in w:
        return ("N=1024 vPBS as IVC chain (BASELINE config 2; verified_pbs ivc_based_vpbs.rs:159-386) via vpbs_ivc_prove_pbs; "
                "step = 1 chained proof of the cyclic circuit (2^16 rows) per chain")
    return _short(w, 200)


def compact_record(d):
    """the one line the driver parses, from the full result `d` of a bench run (rank 0's dict)"""
    r = d.get("roofline") or {}
    c = d.get("cpu_baseline")
    cfg = d.get("config", {})
    mode = str(cfg.get("parallelism", ""))
    line = {
        "metric": METRIC, "value": _num(d.get("value")), "unit": d.get("unit"), "n_gpus": d.get("n_gpus"), "steps": d.get("steps"),
        "warmup": d.get("warmup"), "ms_per_step": _num(d.get("ms_per_step")), "higher_is_better": True, "scaling": d.get("scaling"),
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload_id(d), "chains_per_gpu": cfg.get("chains_per_gpu"),
                   "parallelism": mode.split(":")[0] if mode else None,
                   "witness": ("device+late" if "late phase there as well" in str(cfg.get("early_witness_phase", "")) else "device")
                              if str(cfg.get("early_witness_phase", "")).startswith(("on the device", "device")) else "host"},
        "roofline": {"bound": r.get("bound"), "kernel": str(r.get("kernel", "")).split(" ")[0] or None,
                     "achieved": _num(r.get("achieved")), "peak": _num(r.get("peak")), "unit": r.get("unit"), "frac": _num(r.get("frac")),
                     "traffic": _num(r.get("traffic"), 9),
                     "algorithmic_bytes_per_launch": _num(r.get("algorithmic_bytes_per_launch"), 12), "launch_ms_avg": _num(r.get("launch_ms_avg")),
                     "launches": r.get("launches"), "int_issue_frac": _num(r.get("int_issue_frac")),
                     "valu_budget_frac": _num(_get(r, "valu_budget", "frac")),
                     "counters_fresh": _get(r, "valu_budget", "counters_match_kernel_sources"),
                     "valu_budget_frac_single_chain": _num(_get(r, "valu_budget_single_chain", "frac")),
                     "step_hbm_frac": _num(r.get("step_hbm_frac")), "sclk_mhz": _num(r.get("shader_clock_mhz_in_kernel"), 5)},
        "ms_per_step_proof": _num(d.get("ms_per_step_proof")), "step_proofs_per_s": _num(d.get("step_proofs_per_s")),
        "steps_per_vpbs_proof": d.get("steps_per_vpbs_proof"),
    }
    if c is not None:
        line["cpu_baseline"] = {"value": _num(c.get("value")), "unit": c.get("unit"), "ms_per_step": _num(c.get("ms_per_step")),
                                "cores": c.get("cores"), "kind": c.get("kind"), "runs": c.get("runs"),
                                "sample": _short(c.get("sample_id") or "%s complete synthetic step proofs (2^16 rows, 135 wires), median; "
                                                 "C oracle, OpenMP" % c.get("runs"), 120),
                                "cargo": _get(c, "reference_probe", "cargo")}
    if d.get("rccl") is not None:
        line["rccl"] = {"ranks": _get(d, "rccl", "ranks"), "version": _get(d, "rccl", "version"), "backend": _get(d, "rccl", "backend")}
    line["cpus_per_rank"] = d.get("cpus_per_rank")
    line["launched_by"] = _short(d.get("launched_by", ""), 40) or None
    if d.get("sustained"):
        s = d["sustained"]
        line["sustained"] = {"vpbs_proofs_per_s": _num(s.get("vpbs_proofs_per_s")), "chains": s.get("chains"), "seconds": _num(s.get("seconds")),
                             "over_burst": _num(s.get("sustained_over_burst"))}
    extra = {
        "ivc_chain_seconds": _get(d, "ivc_chain", "seconds"),
        "ivc_chain_ms_per_step": _get(d, "ivc_chain", "ms_per_step"),
        "ivc_chain_decrypted": (_get(d, "ivc_chain", "decrypted") == _get(d, "ivc_chain", "message")) if d.get("ivc_chain") and "error" not in d["ivc_chain"] else None,
        "ivc_single_chain_ms_per_step": _get(d, "ivc_single_chain", "ms_per_step"),
        "ivc_n2048_ms_per_step": _get(d, "ivc_chain_n2048", "ms_per_step"),
        "whole_pbs_seconds": _get(d, "whole_pbs", "seconds"),
        "batch128_step_proofs_per_s": _get(d, "batch_of_128", "step_proofs_per_s"),
        "step_micro_ms": _get(d, "step_micro", "ms_per_step_proof"),
        "sharded_ms_per_step": _get(d, "sharded", "ms_per_step"),
    }
    for k, v in extra.items():
        if v is not None:
            line[k] = _num(v)
    if "parity_checked_full_size" in d:
        line["parity_checked_full_size"] = bool(d["parity_checked_full_size"])
    errors = sorted(k for k, v in d.items() if isinstance(v, dict) and "error" in v)
    if errors:
        line["leg_errors"] = errors
    line["detail"] = "bench_detail.json"
    return _bounded(line)


_LONG = {"workload": 200, "sample": 120, "metric": 200}


def _bounded(x, key=None):
    """identifiers stay identifiers: every string of the line is cut to 48 characters (workload 200, sample 120), whatever a leg put there"""
    if isinstance(x, dict):
        return {k: _bounded(v, k) for k, v in x.items()}
    if isinstance(x, list):
        return [_bounded(v, key) for v in x[:16]]
    if isinstance(x, str):
        return _short(x, _LONG.get(key, 48))
    return x


def dumps(line):
    s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:
        raise ValueError("bench line is %d bytes (limit %d): it is a record, move prose to the detail file" % (len(s), LINE_LIMIT))
    if not s.startswith('{"metric"'):
        raise ValueError("the line must start with {\"metric\" (launch_ranks relays on that prefix)")
    return s


def write_detail(d, root=ROOT, path=None):
    """the full result: to `path` when one is given (bench.py --detail), else next to bench.py and under gpurun_out/ where that exists (the only
    directory that travels back from a GPU box)"""
    if path == os.devnull:
        return []
    if path:
        paths = [path]
    else:
        paths = [os.path.join(root, "bench_detail.json")]
        if os.path.isdir(os.path.join(root, "gpurun_out")):
            paths.append(os.path.join(root, "gpurun_out", "bench_detail.json"))
    written = []
    for p in paths:
        try:
            with open(p + ".tmp", "w") as f:
                json.dump(d, f, indent=1, default=str)
            os.replace(p + ".tmp", p)
            written.append(p)
        except OSError:
            pass
    return written


if __name__ == "__main__":
    import sys
    print(dumps(compact_record(json.load(open(sys.argv[1])))))
