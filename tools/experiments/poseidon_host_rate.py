import sys, time
sys.path.insert(0,'.')
import numpy as np, vpbs_amd
from vpbs_amd import api
rng=np.random.default_rng(1)
big=rng.integers(0,api.P,size=(400000,12),dtype=np.uint64)
for _ in range(3):
    b=big.copy(); t=time.perf_counter(); rc=api.lib().vpbs_k_poseidon_host(api._ptr(b), b.shape[0]); dt=time.perf_counter()-t
    print('path', rc, 'us per permutation (batched x8)', dt/b.shape[0]*1e6)
x=np.arange(8*200000,dtype=np.uint64)
for _ in range(2):
    t=time.perf_counter(); api.hash_no_pad(x); dt=time.perf_counter()-t
    print('us per permutation (scalar chain)', dt/200000*1e6)
