#!/usr/bin/env python3
"""Repeated vpbs_ivc_prove_pbs on one object: resident set size and device memory before / after (a leak in the chain driver, the witness
states or the provers would show as growth per chain)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vpbs_amd  # noqa: E402
from vpbs_amd import api, circuit_file  # noqa: E402


def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


N, K, ELL, LOGB, n_lwe, log_n = 8, 2, 4, 5, 6, 13
cyc, dum = (circuit_file.load(p) for p in circuit_file.find_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n))
c = vpbs_amd.Context(0, log_n_max=16)
ivc = api.Ivc(c, cyc, dum, N, K, K * ELL * K * N)
if len(sys.argv) > 1:   # the device witness pipeline: batch size [late-on-device flag]
    ivc.set_device_witness(ELL, LOGB, int(sys.argv[1]), len(sys.argv) > 2)
keys = c.keygen(N, K, ELL, LOGB, n_lwe, 1, 4.99027217501041e-8, 1.17021618159313e-5)
testv, delta = api.testv(N, 2)
ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta % api.P)
for round_ in range(4):
    for _ in range(50):
        ivc.prove_pbs(testv, ct, keys["bsk"], keys["ksk"])
    free, total = torch.cuda.mem_get_info()
    print("after %3d chains: RSS %.0f MB, device memory in use %.0f MB" % (50 * (round_ + 1), rss_mb(), (total - free) / 1e6))
