#!/usr/bin/env python3
"""One whole verifiable PBS at the paper's parameters on one MI355X, measured end to end instead of extrapolated from the step rate:
n + 2 = 730 chained steps of the reference's step circuit WITHOUT its recursive verifier (the exported circuit description,
verifiable-fhe-paper_amd/circuit_file.py: this tool imports no circuit builder; the hand-over of
accumulator, counter and hash chains between steps is done by this driver, which is what the in-circuit verifier enforces in the
reference).  Pipeline: native accumulator chain on the device (vpbs_pbs_accumulator_chain) and native hash chains on the host -> the
PartialWitness values of every step (the hash chains computed by a host thread beside the device) -> device witness generation in batches (vpbs_witness_device_*) -> gather -> step proofs on
`provers` contexts -> every proof verified on the host (after the clock).  Keys, test vector and the LWE input are the seeded ones of
vpbs_keygen / vpbs_lwe_encrypt / vpbs_testv at the paper's noise levels (main.rs:40-52); the final accumulator -- a public input of the last
proof -- decrypts to the encrypted message under the partial key (main.rs:58-64).
usage: tools/prove_pbs.py [n_lwe=728] [batch=73] [provers=5]  ->  one JSON line
Several GPUs: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/prove_pbs.py ...
  the steps of the ONE PBS are split into N contiguous ranges, one per rank / GPU (they are independent once the accumulator and hash
  chains are known; every rank recomputes those chains -- 40 ms on its device, 2 s on one host core, beside the proving).  No data-path
  collective: a barrier and a max over the ranks' times.  VPBS_PBS_BACKEND=gloo and VPBS_PBS_DEVICE=0 put all ranks on one GPU (tests)."""
import json
import os
import queue
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vpbs_amd  # noqa: E402
from vpbs_amd import api, circuit_file  # noqa: E402

N, K, ELL, LOGB = 1024, 2, 4, 5
P = api.P


def main():
    n_lwe = int(sys.argv[1]) if len(sys.argv) > 1 else 728
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 73
    provers = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    steps = n_lwe + 2
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = int(os.environ.get("VPBS_PBS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("VPBS_PBS_BACKEND", "nccl"), timeout=vpbs_amd.sharding.group_timeout())
    torch.cuda.set_device(device)
    my_first, my_end = steps * rank // world, steps * (rank + 1) // world      # this rank's steps: [my_first, my_end)
    t_all = time.perf_counter()
    b = circuit_file.load(circuit_file.find_step_circuit(N, K, ELL, LOGB, n_lwe))
    sigma = b.circuit.sigma_values()
    targets = b.preset_pos      # acc_init, acc_in, GGSW, counter, mask, the two chain hashes (the exporter's order, ivc_based_vpbs.rs:325-330)
    plan = b.circuit.witness_plan(targets)
    pi_pos = b.pi_pos
    n_constants = b.constants.shape[0]
    cs_values = np.concatenate([b.constants, sigma])
    t_setup = time.perf_counter() - t_all

    ggsw_len = K * ELL * K * N
    main_ctx = vpbs_amd.Context(device, log_n_max=16)
    # main.rs:40-52 with seeded generators: partial key / LWE key, GLWE key, bootstrapping and key-switching keys (on the device), test
    # vector, an LWE encryption of delta * message -- before the clock: key material exists once per key, not per PBS
    t_keys = time.perf_counter()
    message = int(os.environ.get("VPBS_PBS_MESSAGE", "1"))
    keys = main_ctx.keygen(N, K, ELL, LOGB, n_lwe, 0x5EED0728, 4.99027217501041e-8, 1.17021618159313e-5)
    bsk, ksk = keys["bsk"], keys["ksk"]
    testv, delta = api.testv(N, 2)
    ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta * message % P)
    acc_init = np.concatenate([np.zeros((K - 1, N), np.uint64), testv.reshape(1, N)])
    t_keys = time.perf_counter() - t_keys
    ggsws = lambda s: np.zeros(ggsw_len, np.uint64) if s == 0 else (bsk[s - 1] if s <= n_lwe else ksk)
    masks = [int(ct[n_lwe])] + [int(v) for v in ct[:n_lwe]] + [0]
    bsk_h, lwe_h = [np.zeros(4, np.uint64)], [np.zeros(4, np.uint64)]
    hashed = threading.Condition()
    timing = {}
    stop = threading.Event()      # set by the first worker that fails: every wait below polls it, so an error ends the run instead of hanging it

    def fail(e):
        errs.append(e)
        stop.set()
        with hashed:
            hashed.notify_all()

    def q_get(q):
        while not stop.is_set():
            try:
                return q.get(timeout=0.2)
            except queue.Empty:
                continue
        return None

    def hash_thread():
        """the native hash chains (verify_hash_output's sponge over every bootstrapping-key element): sequential by construction, 2.8 ms
        per step on one host core -- runs beside the device, the witness thread waits for the prefix its batch needs"""
        t = time.perf_counter()
        try:
            for s in range(my_end):
                if stop.is_set():
                    return
                hash_one(s)
        except Exception as e:
            fail(e)
        timing["hash"] = time.perf_counter() - t

    def hash_one(s):
        hb = bsk_pre[s + 1] if bsk_pre else api.hash_no_pad(np.concatenate([bsk_h[-1], ggsws(s)]))
        hl = api.hash_no_pad(np.concatenate([lwe_h[-1], np.array([masks[s]], np.uint64)]))
        with hashed:
            bsk_h.append(hb)
            lwe_h.append(hl)
            hashed.notify_all()

    # The bootstrapping-key chain depends on the key only: a deployment computes it once per key.  With several GPUs it is taken as
    # given (otherwise rank r would wait r / N of the 2 s chain before its first witness); on one GPU it stays inside the clock.
    bsk_pre = []
    if os.environ.get("VPBS_PBS_BSK_CHAIN", "precomputed" if world > 1 else "inside") == "precomputed":
        bsk_pre = [np.zeros(4, np.uint64)]
        for s in range(my_end):
            bsk_pre.append(api.hash_no_pad(np.concatenate([bsk_pre[-1], ggsws(s)])))

    def values(first, count):
        v = np.zeros((len(targets), count), np.uint64)
        for j in range(count):
            s = first + j
            acc_in = acc_init if s == 0 else accs[s - 1]
            v[:, j] = np.concatenate([acc_init.reshape(-1), acc_in.reshape(-1), ggsws(s), np.array([s + 1, masks[s]], np.uint64), bsk_h[s], lwe_h[s]])
        return v

    wctx = [vpbs_amd.Context(device, log_n_max=16) for _ in range(2)]
    wdev = [api.WitnessDevice(c, plan, batch) for c in wctx]
    pctx = [vpbs_amd.Context(device, log_n_max=16) for _ in range(provers)]
    if "VPBS_WIDE_THRESHOLD" not in os.environ:
        for c in pctx:   # provers side by side hide each other's latency: the one-lane Poseidon form down to 2048 nodes (bench.py shares_the_gpu)
            c.set_option("wide_threshold", 2048)
    css = [c.commit_values(cs_values) for c in pctx]
    d_sigma = torch.from_numpy(sigma.view(np.int64)).cuda()
    d_wires = [torch.zeros((135, b.n), dtype=torch.int64, device="cuda") for _ in range(provers)]
    digest = np.array([11, 22, 33, 44], np.uint64)
    free_obj, ready, errs = queue.Queue(), queue.Queue(), []
    results = [None] * steps
    for k in range(2):
        free_obj.put(k)
    outstanding, lock = [0, 0], threading.Lock()
    wit_s = []

    def witness_thread():
        try:
            for first in range(my_first, my_end, batch):
                count = min(batch, my_end - first)
                with hashed:
                    while not hashed.wait_for(lambda: stop.is_set() or len(bsk_h) > first + count - 1, timeout=0.5):
                        pass
                if stop.is_set():
                    break
                vals = values(first, count)
                k = q_get(free_obj)
                if k is None:
                    break
                t = time.perf_counter()
                wdev[k].run(vals)
                wit_s.append(time.perf_counter() - t)
                with lock:
                    outstanding[k] = count
                for i in range(count):
                    ready.put((k, i, first + i))
        except Exception as e:
            fail(e)
        for _ in range(provers):
            ready.put(None)

    def prover(j):
        try:
            while True:
                item = q_get(ready)
                if item is None:
                    return
                k, i, s = item
                try:
                    wdev[k].wires(i, d_wires[j].data_ptr())
                    pis = wdev[k].read(i, pi_pos)
                finally:      # the witness object goes back whether or not this instance could be read
                    with lock:
                        outstanding[k] -= 1
                        if outstanding[k] == 0:
                            free_obj.put(k)
                si = pctx[j].make_step_inputs(b.log_n, d_wires[j].data_ptr(), None, None, css[j], digest, pis, on_device=True, shapes=(135, 20, 16),
                                              sigmas=int(d_sigma.data_ptr()), n_routed=80, n_constants=n_constants, gates=b.gates)
                results[s] = (pctx[j].prove_step(si), pis)
        except Exception as e:
            fail(e)

    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    hasher = threading.Thread(target=hash_thread)
    hasher.start()
    accs = main_ctx.pbs_accumulator_chain(acc_init, ct, bsk, ksk, K, ELL, LOGB)            # [steps][K][N]: every step's accumulator
    t_chain = time.perf_counter() - t0
    ts = [threading.Thread(target=witness_thread)] + [threading.Thread(target=prover, args=(j,)) for j in range(provers)]
    for t in ts:
        t.start()
    for t in ts + [hasher]:
        t.join()
    t_mine = time.perf_counter() - t0
    if errs:        # before any collective and before the chain-equality assertions
        raise errs[0]
    if dist:
        dist.barrier()
    t_prove = time.perf_counter() - t0          # all ranks done
    t_hash = timing["hash"]
    # the chain the proofs expose is the native one
    for s in range(my_first, my_end):
        pis = results[s][1]
        assert int(pis[K * N]) == s + 1
        assert (pis[K * N + 1:2 * K * N + 1] == accs[s].reshape(-1)).all(), s
        assert (pis[-8:-4] == bsk_h[s + 1]).all() and (pis[-4:] == lwe_h[s + 1]).all(), s
    # ... and the PBS did its job: the last accumulator decrypts to the message under the partial key (main.rs:58-64)
    m_bar = main_ctx.glwe_decrypt(keys["s_to"], accs[-1])
    decrypted = round(int(m_bar[0]) / delta) % 4
    assert decrypted == message, (decrypted, message)
    t0 = time.perf_counter()
    from concurrent.futures import ThreadPoolExecutor
    cs_cap = css[0].cap()

    def verify(s):
        proof, pis = results[s]
        return api.verify_step(proof, cs_cap, [n_constants + 80, 135, 20, 16], digest, pis, b.log_n, check_permutation=True,
                               n_constants=n_constants, n_routed=80, gates=b.gates)

    with ThreadPoolExecutor(max_workers=8) as pool:      # the host verifier releases the GIL: every proof of the chain is checked
        verdicts = list(pool.map(verify, range(my_first, my_end)))
    assert all(verdicts), [my_first + s for s, v in enumerate(verdicts) if not v][:5]
    checked = my_end - my_first
    t_verify = time.perf_counter() - t0
    if dist:
        t = torch.tensor([t_prove, float(checked)], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        tmax, tsum = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        t_prove, checked = float(tmax[0]), int(tsum[1])
        if rank != 0:
            dist.barrier()
            dist.destroy_process_group()
            return
    proof_bytes = sum(results[my_first][0][k].nbytes for k in ("caps", "openings", "fri"))
    print(json.dumps({
        "what": "one whole vPBS at N=1024, k=1, ELL=4, LOGB=5, n=%d: %d chained step proofs of build_step_circuit (no recursive verifier; "
                "%d gate rows, degree 2^%d) on %d x MI355X%s" % (n_lwe, steps, b.used_rows, b.log_n, world,
                                                           "" if world == 1 else " (the steps split into %d contiguous ranges, one per GPU)" % world),
        "n_gpus": world,
        "step_proofs": steps, "seconds": t_prove, "vpbs_proofs_per_s": 1.0 / t_prove, "step_proofs_per_s": steps / t_prove,
        "ms_per_step_proof": 1e3 * t_prove / steps, "witness_batch": batch, "provers": provers,
        "device_witness_s_per_batch": sum(wit_s) / len(wit_s),
        "bsk_hash_chain": "precomputed per key (before the clock)" if bsk_pre else "inside the clock",
        "inside_the_clock": {"accumulator_chain_on_device_s": t_chain, "native_hash_chains_on_one_host_core_s": t_hash,
                             "note": "the hash chains run beside the device; every witness batch waits for the prefix it needs"},
        "before_the_clock": {"circuit_file_sigma_plan_s": t_setup, "seeded_keygen_on_device_s": t_keys,
                             "note": "once per circuit / once per key, not per PBS"},
        "message": message, "decrypted": decrypted,
        "checks": "the bootstrapped ciphertext decrypts to the message; accumulator / counter / hash public inputs of all %d proofs equal "
                  "the native chains; all %d proofs verified by "
                  "vpbs_verify_step on 8 host threads per rank in %.2f s (after the clock)" % (checked, checked, t_verify),
        "proof_words_kB": proof_bytes / 1e3}))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
