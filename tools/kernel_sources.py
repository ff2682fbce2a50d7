#!/usr/bin/env python3
"""Fingerprint of the device code the counter files under profiles/ were measured on.

bench.py prices the whole step with wave-level instruction counts and HBM bytes that are CONSTANTS read from profiles/ (counter passes are their
own rocprofv3 runs, not part of a bench run).  A kernel edit without a counter refresh would misprice `valu_budget` and `roofline.traffic`
silently: tools/refresh_profiles.sh records this fingerprint next to the counters (profiles/rNN_pmc_sources.json) and
tests/test_host_cpu.py recomputes it -- the CPU suite fails until the counters are measured again on the code as it stands.

usage: python tools/kernel_sources.py  ->  one JSON object {"sha256": ..., "files": {name: sha256}}"""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "verifiable-fhe-paper_amd", "csrc")
# what compiles into the kernels of a step proof (host-only sources -- the witness plans, the IVC driver, the verifier, the communicator -- are not in it)
DEVICE_SOURCES = ["gl.h", "poseidon.h", "poseidon_constants.inc", "poseidon_partial_groups.inc", "kernels.h", "gates.h", "hash.hip", "ntt.hip", "fri.hip",
                  "permutation.hip", "quotient.hip", "gates.hip"]


def fingerprint():
    files = {}
    for name in DEVICE_SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            files[name] = hashlib.sha256(f.read()).hexdigest()
    total = hashlib.sha256("".join("%s:%s\n" % kv for kv in sorted(files.items())).encode()).hexdigest()
    return {"sha256": total, "files": files}


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round NN, or None"""
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return hits[-1] if hits else None


if __name__ == "__main__":
    print(json.dumps(fingerprint(), indent=1))
