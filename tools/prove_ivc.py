#!/usr/bin/env python3
"""One verifiable PBS as the reference produces it: an IVC chain (/root/reference/src/vtfhe/ivc_based_vpbs.rs:159-386 `verified_pbs`) --
n + 2 step proofs of the CYCLIC step circuit, each verifying its predecessor in circuit, so that the LAST proof alone (~190 kB) attests to the
whole bootstrap -- followed by `verify_pbs` (:388-489) on that one proof.

The circuit arrives as data (verifiable-fhe-paper_amd/circuit_file.py: the exported cyclic circuit and its dummy circuit; this tool imports
no circuit builder).  Per step: the PartialWitness (previous proof's words and public inputs, condition bit, GGSW, mask, verifier data) ->
compiled witness generation on the host in two phases (what does not depend on the previous proof is generated ahead on a second thread;
the recursive-verifier rows wait for the proof, so the chain itself is sequential by construction) -> wires to the device -> vpbs_prove_step -> the proof's words feed the next step.  Keys, test vector and the LWE input are the
seeded ones of vpbs_keygen / vpbs_testv / vpbs_lwe_encrypt at the paper's noise.

usage: tools/prove_ivc.py [N=1024] [n_lwe=728] [log_n=16] [steps=all]   ->  one JSON line
  steps < n + 2 proves only a prefix of the chain (tests); verify_pbs's counter / hash checks are then made against that prefix.
  VPBS_IVC_CHAINS=c: c independent PBS (own keys, message, context, plans) side by side on the one GPU: a chain leaves the GPU idle while its
  host phases run, a second chain fills those gaps -- throughput, not latency.
Several GPUs (BASELINE config 4): python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/prove_ivc.py ...
  the chain is sequential, so the GPUs share every STEP: each step proof is coset-sharded over the ranks (vpbs_prove_step_sharded: a rank
  computes the LDEs, leaf hashes and Merkle subtrees of its cosets; cap hashes, quotient values and query records travel over the library's
  RCCL collectives), every rank generates the (identical) witness on its host and ends with the identical proof.  VPBS_PBS_BACKEND=gloo and
  VPBS_PBS_DEVICE=0 put all ranks on one GPU with the callback communicator (tests)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
# several chains = many streams: see bench.py (16 hardware queues for the device witness pipeline, 8 for the host pipeline)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if os.environ.get("VPBS_IVC_DEVICE_WITNESS", "0") not in ("", "0") else "8")
import torch  # noqa: E402

import vpbs_amd  # noqa: E402
from vpbs_amd import api, circuit_file  # noqa: E402

K, ELL, LOGB = 2, 4, 5
P = api.P


class Circuit:
    """one exported circuit on one context: constants/sigmas commitment, verifier data, witness plan, prove / verify"""

    def __init__(self, ctx, path, comm=None, dist_device=None):
        self.ctx, self.comm = ctx, comm
        self.d = d = circuit_file.load(path)
        self.sigma = d.circuit.sigma_values()
        self.cs_values = np.concatenate([d.constants, self.sigma])
        if comm is None:
            self.cs = ctx.commit_values(self.cs_values)
            self.cap = self.cs.cap()
        else:   # this rank's cosets of the constants/sigmas commitment; the cap is assembled from all ranks
            from vpbs_amd import sharding
            self.d_cs = torch.from_numpy(self.cs_values.view(np.int64)).cuda()
            torch.cuda.synchronize()
            self.cs, self.cap = sharding.sharded_commit(ctx, self.d_cs.data_ptr(), self.cs_values.shape[0], d.log_n, device=dist_device)
            self.cap = self.cap.reshape(-1, 4)
        self.vk = circuit_file.verifier_data_words(self.cap, d.log_n)
        self.d_sigma = torch.from_numpy(self.sigma.view(np.int64)).cuda()
        self.plan = d.circuit.witness_plan(d.preset_pos)
        pi = np.array(d.pi_pos)
        self.pi_cols, self.pi_rows = pi[:, 0], pi[:, 1]
        self.ncols = [d.n_constants + 80, 135, 20, 16]

    def prove(self, d_wires_ptr, pis):
        si = self.ctx.make_step_inputs(self.d.log_n, d_wires_ptr, None, None, self.cs, self.vk[:4], pis, on_device=True, shapes=(135, 20, 16),
                                       sigmas=int(self.d_sigma.data_ptr()), n_routed=80, n_constants=self.d.n_constants, gates=self.d.gates)
        return self.ctx.prove_step(si, self.comm), si

    def verify(self, proof, pis):
        return api.verify_step(proof, self.cap, self.ncols, self.vk[:4], pis, self.d.log_n, n_constants=self.d.n_constants, n_routed=80,
                               gates=self.d.gates)


def run_chain(ctx, cyc, dum, N, n_lwe, log_n, steps, seed, message, start, dist=None):
    """one vPBS: keys from `seed`, base proof, `steps` chained proofs, verify_pbs on the last -> result dict.  `start`: a threading.Barrier
    shared by the chains of this process (several independent PBS chains keep one GPU busy while each waits for its host phases)."""
    import queue
    import threading
    total = n_lwe + 2
    shape_words = cyc.d.meta["proof_words"]
    n_pi, kn = len(cyc.d.pi_pos), K * N
    assert n_pi == 2 * kn + 9 + 68 and len(dum.d.preset_pos) == n_pi
    # main.rs:40-52 with seeded generators (before the clock: key material exists once per key)
    t_keys = time.perf_counter()
    keys = ctx.keygen(N, K, ELL, LOGB, n_lwe, seed, 4.99027217501041e-8, 1.17021618159313e-5)
    testv, delta = api.testv(N, 2)
    ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta * message % P)
    acc_init = np.concatenate([np.zeros((K - 1, N), np.uint64), testv.reshape(1, N)])
    t_keys = time.perf_counter() - t_keys
    ggsw_len = K * ELL * K * N
    zero_ggsw = np.zeros(ggsw_len, np.uint64)
    plan_steps = [(0, zero_ggsw, int(ct[n_lwe]))] + [(1, keys["bsk"][x], int(ct[x])) for x in range(n_lwe)] + [(1, keys["ksk"], 0)]

    n_buf = 3
    bufs = [torch.empty((135, cyc.d.n), dtype=torch.int64).pin_memory() for _ in range(n_buf)]
    views = [b.numpy().view(np.uint64) for b in bufs]
    d_bufs = [torch.empty((135, cyc.d.n), dtype=torch.int64, device="cuda") for _ in range(n_buf)]
    flat = lambda p: np.concatenate([np.asarray(p[k], np.uint64).reshape(-1) for k in ("caps", "openings", "fri")])
    # The previous proof is the LATE part of a step's PartialWitness: everything that does not depend on it -- the step logic, both chain
    # hashes, the public-input hashes, i.e. three quarters of the generator work -- is generated ahead by a second host thread
    # (vpbs_witness_plan_split / run_early), which also yields the step's public inputs, the only thing the next early phase needs.  A third
    # thread moves that matrix to the device while the current step is being proven (vpbs_device_upload_bg); once the proof exists, the late
    # phase fills the in-circuit verifier's rows and only those rows are uploaded again (vpbs_device_upload_rows).
    late_mask = np.zeros(len(cyc.d.preset_pos), np.uint8)
    late_mask[:shape_words] = 1
    cyc.plan.split(late_mask)
    free_bufs, generated, ready, errs = queue.Queue(), queue.Queue(), queue.Queue(), []
    late_rows = cyc.plan.late_rows()
    for i in range(n_buf):
        free_bufs.put(i)
    base_pis = np.concatenate([acc_init.reshape(-1), np.zeros(1 + kn + 8, np.uint64), cyc.vk])
    t_early = [0.0]

    def early_thread():
        try:
            pis_prev = base_pis
            zeros = np.zeros(shape_words, np.uint64)
            filled = set()
            for s in range(steps):
                cond, ggsw, mask = plan_steps[s]
                b = free_bufs.get()
                t = time.perf_counter()
                values = np.concatenate([zeros, pis_prev, np.array([cond], np.uint64), ggsw, np.array([mask], np.uint64), cyc.vk, dum.vk,
                                         dummy_flat, np.zeros(base_pis.size, np.uint64)])
                state = cyc.plan.run_early(values, views[b], recycled=b in filled)   # a matrix this plan filled before: values only
                filled.add(b)
                pis_prev = views[b][cyc.pi_cols, cyc.pi_rows].copy()     # public inputs never depend on the inner proof's words
                t_early[0] += time.perf_counter() - t
                generated.put((b, state, values, pis_prev))
        except Exception as e:
            errs.append(e)
            generated.put(None)

    def upload_thread():
        try:
            for s in range(steps):
                item = generated.get()
                if item is None:
                    break
                ctx.upload_bg(d_bufs[item[0]].data_ptr(), bufs[item[0]].data_ptr(), 135 * cyc.d.n)
                ready.put(item)
        except Exception as e:
            errs.append(e)
        if errs:
            ready.put(None)

    # the second proof slot (dummy_proof_and_vk): the dummy circuit's proof of all-zero public inputs, the same in every step -- before the clock
    base_host = torch.empty((135, dum.d.n), dtype=torch.int64).pin_memory()
    zero_pis = np.zeros(base_pis.size, np.uint64)
    dum.plan.run(zero_pis, out=base_host.numpy().view(np.uint64))
    dummy_flat = flat(dum.prove(base_host.cuda().data_ptr(), zero_pis)[0])
    t_wit = t_copy = t_prove = 0.0
    torch.cuda.synchronize()
    start.wait()
    t0 = time.perf_counter()
    worker = threading.Thread(target=early_thread, daemon=True)   # a failure of the main loop must not leave the process waiting on it
    worker.start()
    uploader = threading.Thread(target=upload_thread, daemon=True)
    uploader.start()
    # cyclic_base_proof (ivc_based_vpbs.rs:292-299): a proof of the dummy circuit whose public inputs carry the initial accumulator and the
    # cyclic circuit's verifier data
    dum.plan.run(base_pis, out=base_host.numpy().view(np.uint64))
    d_base = base_host.cuda()
    torch.cuda.synchronize()
    proof, _ = dum.prove(d_base.data_ptr(), base_pis)
    t_base = time.perf_counter() - t0
    last = None
    for s in range(steps):
        item = ready.get()
        if item is None:
            raise errs[0]
        b, state, values, pis = item
        t = time.perf_counter()
        values[:shape_words] = flat(proof)
        cyc.plan.run_late(state, values, views[b])
        t_wit += time.perf_counter() - t
        t = time.perf_counter()
        ctx.upload_rows(d_bufs[b].data_ptr(), bufs[b].data_ptr(), 135, cyc.d.n, *late_rows)
        t_copy += time.perf_counter() - t
        t = time.perf_counter()
        proof, last = cyc.prove(d_bufs[b].data_ptr(), pis)
        t_prove += time.perf_counter() - t
        free_bufs.put(b)
    worker.join()
    uploader.join()
    if dist:
        dist.barrier()
    seconds = time.perf_counter() - t0

    # verify_pbs (:388-489) on the LAST proof only
    t = time.perf_counter()
    blob = ctx.step_proof_to_bytes(last, cyc.d.n_constants, proof)
    back, back_pis = api.step_proof_from_bytes(blob, cyc.ncols, log_n, cyc.d.n_constants)
    assert cyc.verify(back, back_pis), "the final proof does not verify"
    t_verify = time.perf_counter() - t
    if steps == total:   # ... the same through the one-call form of the library (vpbs_verify_pbs), which insists on counter = n + 2
        ok, why = api.verify_pbs(blob, cyc.cap, cyc.ncols, cyc.vk[:4], log_n, cyc.d.n_constants, 80, cyc.d.gates, N, K, testv, ct, keys["bsk"],
                                 keys["ksk"], out_ct=back_pis[kn + 1:2 * kn + 1])
        assert ok, why
    assert (back_pis[:kn] == acc_init.reshape(-1)).all() and int(back_pis[kn]) == steps          # test vector, number of steps
    assert (back_pis[-68:] == cyc.vk).all()                                                      # check_cyclic_proof_verifier_data
    bsk_items = np.stack([zero_ggsw] + [keys["bsk"][x] for x in range(min(steps - 1, n_lwe))] + ([keys["ksk"]] if steps == total else []))
    lwe_items = np.array([[plan_steps[s][2]] for s in range(steps)], np.uint64)
    assert api.hash_chain(bsk_items, back_pis[2 * kn + 1:2 * kn + 5])[1] and api.hash_chain(lwe_items, back_pis[2 * kn + 5:2 * kn + 9])[1]
    accs = ctx.pbs_accumulator_chain(acc_init, ct, keys["bsk"], keys["ksk"], K, ELL, LOGB)
    assert (back_pis[kn + 1:2 * kn + 1] == accs[steps - 1].reshape(-1)).all()                    # the accumulator the native chain reaches
    decrypted = None
    if steps == total:
        m_bar = ctx.glwe_decrypt(keys["s_to"], back_pis[kn + 1:2 * kn + 1].reshape(K, N))
        decrypted = round(int(m_bar[0]) / delta) % 4
        assert decrypted == message, (decrypted, message)
    return {"seconds": seconds, "split": {"witness_late_phase_host": 1e3 * t_wit / steps, "late_rows_to_device": 1e3 * t_copy / steps,
                                          "prove_step": 1e3 * t_prove / steps, "base_proof_once": 1e3 * t_base,
                                          "witness_early_phase_on_a_second_thread": 1e3 * t_early[0] / steps},
            "proof_bytes": len(blob), "verify_ms": 1e3 * t_verify, "message": message, "decrypted": decrypted, "keygen_s": t_keys}


def check_chain(ctx, d, vk, blob, keys, testv, delta, ct, N, n_lwe, log_n, steps, message):
    """verify_pbs (ivc_based_vpbs.rs:388-489) on the LAST proof of a chain of `steps` proofs (bytes as vpbs_ivc_prove_pbs returned them): parse,
    full vpbs_verify_step, test vector, counter, verifier data, the native accumulator chain; the whole chain (steps = n + 2) also through
    vpbs_verify_pbs and decrypted, a prefix against the prefix of both hash chains -> (seconds of the proof verification, decrypted message)"""
    total, kn = n_lwe + 2, K * N
    ncols = [d.n_constants + 80, 135, 20, 16]
    tv = time.perf_counter()
    back, back_pis = api.step_proof_from_bytes(blob, ncols, log_n, d.n_constants)
    assert api.verify_step(back, vk[4:].reshape(-1, 4), ncols, vk[:4], back_pis, log_n, n_constants=d.n_constants, n_routed=80, gates=d.gates), \
        "the final proof does not verify"
    t_verify = time.perf_counter() - tv
    acc_init = np.concatenate([np.zeros((K - 1) * N, np.uint64), testv])
    assert (back_pis[:kn] == acc_init).all() and int(back_pis[kn]) == steps and (back_pis[-68:] == vk).all()
    accs = ctx.pbs_accumulator_chain(acc_init.reshape(K, N), ct, keys["bsk"], keys["ksk"], K, ELL, LOGB)
    assert (back_pis[kn + 1:2 * kn + 1] == accs[steps - 1].reshape(-1)).all()                    # the accumulator the native chain reaches
    decrypted = None
    if steps == total:
        ok, why = api.verify_pbs(blob, vk[4:].reshape(-1, 4), ncols, vk[:4], log_n, d.n_constants, 80, d.gates, N, K, testv, ct, keys["bsk"],
                                 keys["ksk"], out_ct=accs[-1])
        assert ok, why
        m_bar = ctx.glwe_decrypt(keys["s_to"], back_pis[kn + 1:2 * kn + 1].reshape(K, N))
        decrypted = round(int(m_bar[0]) / delta) % 4
        assert decrypted == message, (decrypted, message)
    else:   # a prefix of the chain: the hash chains against the prefix of the keys
        zero = np.zeros(K * ELL * K * N, np.uint64)
        bsk_items = np.stack([zero] + [keys["bsk"][x] for x in range(min(steps - 1, n_lwe))])
        lwe_items = np.array([[int(ct[n_lwe])]] + [[int(ct[x])] for x in range(min(steps - 1, n_lwe))], np.uint64)
        assert api.hash_chain(bsk_items, back_pis[2 * kn + 1:2 * kn + 5])[1] and api.hash_chain(lwe_items, back_pis[2 * kn + 5:2 * kn + 9])[1]
    return t_verify, decrypted


def run_chain_native(ctx, ivc, d, N, n_lwe, log_n, steps, seed, message, start, dist=None):
    """the same PBS through the library's own driver (vpbs_ivc_prove_pbs: the loop of run_chain in C++, csrc/ivc.hip) -> result dict"""
    total, kn = n_lwe + 2, K * N
    t_keys = time.perf_counter()
    keys = ctx.keygen(N, K, ELL, LOGB, n_lwe, seed, 4.99027217501041e-8, 1.17021618159313e-5)
    testv, delta = api.testv(N, 2)
    ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta * message % P)
    t_keys = time.perf_counter() - t_keys
    vk, _ = ivc.verifier_data()
    ncols = [d.n_constants + 80, 135, 20, 16]
    torch.cuda.synchronize()
    start.wait()
    blob, t = ivc.prove_pbs(testv, ct, keys["bsk"], keys["ksk"], steps)
    if dist:
        dist.barrier()
    t_verify, decrypted = check_chain(ctx, d, vk, blob, keys, testv, delta, ct, N, n_lwe, log_n, steps, message)
    return {"seconds": t["seconds"], "split": {"witness_late_phase_host": t["late_witness_ms"], "late_rows_to_device": t["late_rows_upload_ms"],
                                               "prove_step": t["prove_step_ms"], "base_proof_once": t["base_proof_ms"],
                                               "witness_early_phase_on_a_second_thread": t["early_witness_ms"],
                                               "late_stages_run_during_the_previous_proofs_fri_stage": t["late_ahead_ms"]},
            "proof_bytes": len(blob), "verify_ms": 1e3 * t_verify, "message": message, "decrypted": decrypted, "keygen_s": t_keys}


class CpuByRole:
    """VPBS_CPU_BY_ROLE=1: CPU time of this process by thread role while the chains run -- /proc/self/task/*/stat sampled twice a second (threads
    come and go), the library names its threads (vpbs-early-pool, vpbs-late-pool, vpbs-hash, vpbs-early, vpbs-upload, vpbs-late-ahead,
    vpbs-batcher, vpbs-stager), the chains' calling threads are named vpbs-chain here, everything else keeps its own name (HIP runtime
    threads, python).  What a rank's share of the host CPUs is spent on: VERDICT r03 next 7."""

    def __init__(self):
        import threading
        self.seen, self.base, self.stop = {}, {}, threading.Event()
        self.hz = os.sysconf("SC_CLK_TCK")
        self.sample(self.base)
        self.thread = threading.Thread(target=self.loop, daemon=True)
        self.thread.start()

    def sample(self, into):
        for tid in os.listdir("/proc/self/task"):
            try:
                raw = open("/proc/self/task/%s/stat" % tid).read()
            except OSError:
                continue
            name = raw[raw.index("(") + 1:raw.rindex(")")]
            f = raw[raw.rindex(")") + 2:].split()
            into[tid] = (name, (int(f[11]) + int(f[12])) / self.hz)      # utime + stime, seconds

    def loop(self):
        while not self.stop.wait(0.5):
            self.sample(self.seen)

    def report(self, steps_total):
        self.stop.set()
        self.thread.join()
        self.sample(self.seen)
        by_role = {}
        for tid, (name, secs) in self.seen.items():
            secs -= self.base.get(tid, (name, 0.0))[1] if self.base.get(tid, (None,))[0] == name else 0.0
            by_role[name] = by_role.get(name, 0.0) + secs
        total = sum(by_role.values())
        top = sorted(((secs - (self.base.get(tid, (name, 0.0))[1] if self.base.get(tid, (None,))[0] == name else 0.0), name, tid)
                      for tid, (name, secs) in self.seen.items()), reverse=True)[:14]
        return {"cpu_seconds_total": round(total, 3), "cpu_ms_per_chained_step": round(1e3 * total / max(1, steps_total), 3),
                "busiest_threads_cpu_s": [[name, tid, round(secs, 3)] for secs, name, tid in top],
                "cpu_ms_per_chained_step_by_role": {k: round(1e3 * v / max(1, steps_total), 3) for k, v in sorted(by_role.items(), key=lambda kv: -kv[1])
                                                    if v > 0}}


def name_this_thread(name):
    import ctypes
    try:
        ctypes.CDLL(None).prctl(15, name.encode(), 0, 0, 0)      # PR_SET_NAME
    except Exception:
        pass


def main():
    import threading
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n_lwe = int(sys.argv[2]) if len(sys.argv) > 2 else 728
    log_n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    total = n_lwe + 2
    steps = min(total, int(sys.argv[4])) if len(sys.argv) > 4 else total
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = int(os.environ.get("VPBS_PBS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    n_chains = int(os.environ.get("VPBS_IVC_CHAINS", "1")) if world == 1 else 1
    # one chain on a host with the CPUs: 14 threads for the late witness phase (28 independent FRI queries in its last stage); several chains: the default
    api.host_set_late_threads(api.late_threads_for(n_chains, api.host_cpu_budget() // max(1, world)))
    api.host_set_early_threads(api.early_threads_for(n_chains))
    torch.cuda.set_device(device)
    dist, comm, native, dist_device = None, None, False, None
    t_setup = time.perf_counter()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("VPBS_PBS_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device), timeout=vpbs_amd.sharding.group_timeout())
            dist_device = torch.device("cuda", device)
        else:
            dist.init_process_group(backend, timeout=vpbs_amd.sharding.group_timeout())
        dist.barrier()
    cyc_path, dummy_path = circuit_file.find_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n)
    # the chain runs inside the library (vpbs_ivc_prove_pbs; with several GPUs every rank calls it with its communicator);
    # VPBS_IVC_DRIVER=python: the same loop spelled out here over the C ABI
    native_driver = os.environ.get("VPBS_IVC_DRIVER", "library") != "python"
    chains = []
    for ci in range(n_chains):   # every chain has its own context (stream, device memory), circuit commitments and witness plans
        ctx = vpbs_amd.Context(device, log_n_max=max(16, log_n))
        if world > 1:
            from vpbs_amd import sharding
            stage_words = (2 << (log_n + 3)) // world
            native = dist.get_backend() == "nccl" and os.environ.get("VPBS_COMM", "rccl") == "rccl"
            comm = sharding.make_comm_rccl(ctx, stage_words=stage_words) if native else \
                sharding.make_comm(device=dist_device, stage_words=stage_words, stage_device=torch.device("cuda", device))
        if native_driver:
            cd, dd = circuit_file.load(cyc_path), circuit_file.load(dummy_path)
            if n_chains > 1 and "VPBS_WIDE_THRESHOLD" not in os.environ:
                ctx.set_option("wide_threshold", 2048)   # chains side by side hide latency: the one-lane Poseidon form down to 2048 nodes (bench.py)
            ivc = api.Ivc(ctx, cd, dd, N, K, K * ELL * K * N, comm)
            if int(os.environ.get("VPBS_IVC_DEVICE_WITNESS", "0")):   # early witness phases on the device, this many steps per batch
                ivc.set_device_witness(ELL, LOGB, int(os.environ["VPBS_IVC_DEVICE_WITNESS"]), os.environ.get("VPBS_IVC_DEVICE_LATE", "") not in ("", "0"))
            chains.append((ctx, ivc, cd))
        else:
            chains.append((ctx, Circuit(ctx, cyc_path, comm, dist_device), Circuit(ctx, dummy_path)))
    t_setup = time.perf_counter() - t_setup
    message = int(os.environ.get("VPBS_PBS_MESSAGE", "1"))
    start = threading.Barrier(n_chains)
    results, errors = [None] * n_chains, []

    def chain_thread(ci):
        try:
            name_this_thread("vpbs-chain")
            torch.cuda.set_device(device)
            ctx, cyc, dum = chains[ci]
            if native_driver:
                results[ci] = run_chain_native(ctx, cyc, dum, N, n_lwe, log_n, steps, 0x5EED0728 + ci, (message + ci) % 2, start, dist)
            else:
                results[ci] = run_chain(ctx, cyc, dum, N, n_lwe, log_n, steps, 0x5EED0728 + ci, (message + ci) % 2, start, dist)
        except BaseException as e:                           # noqa: BLE001
            errors.append(e)
            start.abort()

    cpu_by_role = CpuByRole() if os.environ.get("VPBS_CPU_BY_ROLE", "") not in ("", "0") else None
    t_all = time.perf_counter()
    # every chain on a thread of its own, the main thread only waits: the interpreter's main thread is not a good place for a chain
    # (measured at 2 CPUs: the chain on the main thread used 0.85 of a CPU, the others 0.2 each)
    threads = [threading.Thread(target=chain_thread, args=(ci,)) for ci in range(n_chains)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    t_all = time.perf_counter() - t_all
    cpu_report = cpu_by_role.report(steps * n_chains) if cpu_by_role else None   # (includes each chain's verification after its clock)
    if errors:
        raise errors[0]
    ctx, cyc, dum = chains[0]
    r0 = results[0]
    seconds = max(r["seconds"] for r in results)
    desc = dum if native_driver else cyc.d
    n_pi = len(desc.pi_pos)
    if rank != 0:
        if native_driver:
            cyc.free()               # the communicator must outlive the vpbs_ivc
        if native:
            sharding.free_comm_rccl(comm)
        dist.barrier()
        dist.destroy_process_group()
        return
    print(json.dumps({
        "n_gpus": world, "step_proofs_sharded": ("every step proof coset-sharded over %d GPUs (%s)" % (world, "native RCCL" if native else
                                                 dist.get_backend())) if world > 1 else None,
        "what": "%s as an IVC chain (ivc_based_vpbs.rs verified_pbs): %d of the %d step proofs of the CYCLIC step circuit (step logic + "
                "in-circuit verifier of the previous proof: %d gate rows, degree 2^%d, %d public inputs) at N=%d, k=1, ELL=4, LOGB=5, n=%d on "
                "%d x MI355X; the last proof alone is the vPBS proof" % ("one vPBS" if n_chains == 1 else "%d independent vPBS side by side, each" %
                                                                         n_chains, steps, total, desc.meta.get("used_rows", 0), log_n, n_pi, N, n_lwe, world),
        "host": {"cpus_in_affinity_mask": len(os.sched_getaffinity(0)), "cgroup_cpu_max": (open("/sys/fs/cgroup/cpu.max").read().strip()
                                                                                              if os.path.exists("/sys/fs/cgroup/cpu.max") else None),
                 "loadavg": os.getloadavg()[0]},
        "chains": n_chains, "driver": "vpbs_ivc_prove_pbs (csrc/ivc.hip)" if native_driver else "the loop of tools/prove_ivc.py over the C ABI",
        "step_proofs": steps, "seconds": seconds,
        "seconds_full_chain_extrapolated": None if steps == total else seconds / steps * total,
        "vpbs_proofs_per_s": (n_chains / seconds) if steps == total else None, "ms_per_step": 1e3 * seconds / steps,
        "ms_per_step_split": r0["split"], "ms_per_step_split_other_chains": [r["split"] for r in results[1:]],
        "proof_bytes": r0["proof_bytes"], "verify_last_proof_ms": r0["verify_ms"], "message": r0["message"], "decrypted": r0["decrypted"],
        "other_chains": [{k: r[k] for k in ("seconds", "message", "decrypted")} for r in results[1:]],
        "before_the_clock": {"circuit_files_commit_plan_s": t_setup, "seeded_keygen_s": r0["keygen_s"]},
        "cpu_by_role": cpu_report,
        "checks": "final proof serialised, parsed back and verified by vpbs_verify_step (full check); its public inputs carry the test vector, "
                  "counter = number of steps, the circuit's own verifier data, the native accumulator and both native chain hashes"
                  + ("; the bootstrapped ciphertext decrypts to the message" if steps == total else "")}))
    for ctx, cyc, dum in chains:
        if native_driver:
            cyc.free()
            continue
        cyc.plan.free()
        dum.plan.free()
        cyc.cs.free()
        dum.cs.free()
    if native:
        sharding.free_comm_rccl(comm)
    for ctx, _, _ in chains:
        ctx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
