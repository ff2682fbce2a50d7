import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + '/tests', ROOT + '/circuitgen']
import numpy as np
import vpbs_amd
from vpbs_amd import api, circuit_file
import tfhe_oracle as T
from test_cyclic_cpu import n8_chain_inputs
P = api.P
N, K, ELL, LOGB, n_lwe, log_n = 8, 2, 4, 5, 1, 13
ring, (s_to, s_lwe, s_glwe, bsk, ksk), delta, testv, ct = n8_chain_inputs()
cyc, dum = (circuit_file.load(p) for p in circuit_file.find_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n))
W, n_pi = cyc.meta["proof_words"], len(cyc.pi_pos)
kn = K * N
plan = cyc.circuit.witness_plan(cyc.preset_pos)
late = np.zeros(len(cyc.preset_pos), np.uint8); late[:W] = 1
plan.split(late)
acc_init = [[0] * N for _ in range(K - 1)] + [list(testv)]
accs = T.pbs_chain(ring, acc_init, ct, bsk, ksk, K, ELL, LOGB)
g = K * ELL * K * N
ggsws = [np.zeros(g, np.uint64)] + [T.flatten_ggsw(x) for x in bsk] + [T.flatten_ggsw(ksk)]
masks = [int(ct[n_lwe])] + [int(ct[x]) for x in range(n_lwe)] + [0]
vk = np.arange(68, dtype=np.uint64) + 11
flat = lambda acc: [int(v) for p in acc for v in p]
pis = [np.array(flat(acc_init) + [0] + [0] * kn + [0] * 8 + [int(x) for x in vk], np.uint64)]
hb = np.zeros(4, np.uint64); hl = np.zeros(4, np.uint64)
for s in range(n_lwe + 2):
    hb = api.hash_no_pad(np.concatenate([hb, ggsws[s]]))
    hl = api.hash_no_pad(np.concatenate([hl, np.array([masks[s]], np.uint64)]))
    pis.append(np.array(flat(acc_init) + [s + 1] + flat(accs[s]) + [int(x) for x in hb] + [int(x) for x in hl] + [int(x) for x in vk], np.uint64))
rng = np.random.default_rng(1)
dvk = rng.integers(0, int(P), 68, dtype=np.uint64); dproof = rng.integers(0, int(P), W, dtype=np.uint64)
u = lambda x: np.array([x], np.uint64)
ctx = vpbs_amd.Context(0, log_n_max=16)
dev = api.WitnessDevice(ctx, plan, max_batch=4, early=True)
def col(s, proof=None, zero_ggsw=None):
    return np.concatenate([np.zeros(W, np.uint64) if proof is None else proof, pis[s], u(0 if s == 0 else 1), ggsws[s] if zero_ggsw is None else zero_ggsw,
                           u(masks[s] % P), vk, dvk, dproof, np.zeros(n_pi, np.uint64)])
fe = lambda k: rng.integers(0, int(P), k, dtype=np.uint64)
def rnd(cond):
    ip = fe(n_pi); ip[-68:] = vk
    return np.concatenate([fe(W), ip, u(cond), fe(g), fe(1), vk, dvk, dproof, np.zeros(n_pi, np.uint64)])
def with_pis(s, ip):
    return np.concatenate([np.zeros(W, np.uint64), ip, u(0 if s == 0 else 1), ggsws[s], u(masks[s] % P), vk, dvk, dproof, np.zeros(n_pi, np.uint64)])
def variant(s, **kw):
    ip = pis[s].copy()
    if kw.get("counter") is not None: ip[kn] = kw["counter"]
    if kw.get("acc"): ip[kn + 1:2 * kn + 1] = fe(kn)
    if kw.get("hashes"): ip[2 * kn + 1:2 * kn + 9] = fe(8)
    return with_pis(s, ip)
tests = [("step0", [col(0)]), ("random cond 1", [rnd(1)]), ("step1", [col(1)]), ("step0 again", [col(0)]), ("random cond 1 again", [rnd(1)]),
         ("step1 counter 5", [variant(1, counter=5)]), ("step1 random acc", [variant(1, acc=True)]), ("step1 random hashes", [variant(1, hashes=True)]),
         ("step1 cond 0", [np.concatenate([np.zeros(W, np.uint64), pis[1], u(0), ggsws[1], u(masks[1] % P), vk, dvk, dproof, np.zeros(n_pi, np.uint64)])]),
         ("step1 random ggsw", [np.concatenate([np.zeros(W, np.uint64), pis[1], u(1), fe(g), u(masks[1] % P), vk, dvk, dproof, np.zeros(n_pi, np.uint64)])]),
         ("step1 random mask", [np.concatenate([np.zeros(W, np.uint64), pis[1], u(1), ggsws[1], fe(1), vk, dvk, dproof, np.zeros(n_pi, np.uint64)])])]
for name, cols in tests:
    try:
        dev.run(np.ascontiguousarray(np.stack(cols, axis=1)))
        print(name, "ok")
    except Exception as e:
        print(name, "FAILED", str(e)[-60:])
host = np.zeros((135, cyc.n), np.uint64)
try:
    plan.run_early(col(1), host); print("host early step1 ok")
except Exception as e:
    print("host early step1", e)
