"""host-side Poseidon speed of the product library (the Fiat-Shamir transcript hashes ~190 permutations per step proof)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vpbs_amd import api
x = np.arange(1036 * 8, dtype=np.uint64)
api.hash_no_pad(x[:8])
t = time.perf_counter()
for _ in range(20):
    api.hash_no_pad(x)
dt = (time.perf_counter() - t) / 20
print("host poseidon: %.2f us per permutation" % (dt * 1e6 / 1036))
os.system("grep -m1 'model name' /proc/cpuinfo")
