#!/usr/bin/env python3
"""Headline benchmark: vPBS proofs/s at N = 1024 on N GPUs (BASELINE.json metric).

The default workload is the reference's own object: verifiable PBS proofs as IVC chains (`verified_pbs`,
/root/reference/src/vtfhe/ivc_based_vpbs.rs:159-386) driven by the library's vpbs_ivc_prove_pbs.  A "step" = one CHAINED step proof of
the cyclic step circuit (step logic + in-circuit verifier of the previous proof; 2^16 rows, 135 wires) of every chain on the GPU, with
everything a step of the chain costs inside the clock: both witness phases on the host, the uploads, the proof with its Fiat-Shamir
transcript, and the dependency of each proof on the one before.  A vPBS proof = base proof + 730 chained step proofs
(/root/reference/src/main.rs:27, n + 2), so value = chained step proofs/s / 730.  --warmup chained steps run untimed, then exactly --steps
are timed between barrier + synchronise (the clock is placed from the library's progress hook, vpbs_ivc_set_step_callback).

`--workload step` (and the `step_micro` object of the default line) is the synthetic back-to-back step proof over seeded random columns
resident in HBM that rounds 1-2 reported as the headline: no witness generation, no chain dependency, boost clock -- the device-side
prover alone; the full-size parity check against the CPU oracle is made on that instance.

Multi-GPU: independent chains per GPU ("replicas", weak scaling, no data-path collective; SURVEY.md 8e batch mode); --mode sharded: ONE
chain whose every step proof is coset-sharded over the GPUs.
Launch: python bench.py [--gpus N --steps K --warmup W].  N > 1: either under torch.distributed.run (one rank per GPU; RANK / WORLD_SIZE in the
environment), or plainly -- the process then is the PARENT: it touches no GPU and starts torch.distributed.run as a child (launch_ranks).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Several chains per GPU mean 20-30 HIP streams (per chain: the prover's, its upload stream, the witness contexts').  The HIP runtime maps a
# process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise: measured with six chains per
# GPU (tools/experiments/hwq_ab.sh), 4 / 8 / 16 / 24 queues = 8.45-8.54 / 8.35-8.40 / 8.69-8.75 / 8.73-8.78 ms per chained proof with the host
# witness pipeline and 8.72-8.74 / 8.32-8.35 / 8.31-8.32 / 8.58-8.66 with the device pipeline; with the waits on completion words (round 4) the
# device pipeline -- a chain there has six streams -- gains from 16: 8.70 -> 8.48 on 16 CPUs, 9.49 -> 9.21 on 4 (tools/experiments/ivc_matrix.sh --preset hw_queues_dw),
# the host pipeline loses (six chains 8.17 -> 8.38; eight chains 7.94-8.10 / 8.08-8.11 / 8.17-8.21 with 8 / 12 / 16: tools/experiments/ivc_matrix.sh --preset host8_queues).
# Read by the runtime when it starts: decided here, from the CPU share that later picks the pipeline.
def _cpu_share_before_hip():
    """this rank's share of the CPUs the container may use, without touching the library (the HIP runtime reads its environment when it starts):
    affinity mask and cgroup CPU quota, divided among the ranks of the node -- the figure vpbs_host_cpu_budget() / world gives later"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))))


def visible_gpus(top="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process could open, counted WITHOUT opening one (the launching parent must never initialise the device): the KFD topology
    under /sys (a node with simd_count > 0 is a GPU), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.
    Returns None where the topology is not readable -- the caller then lets the ranks find out."""
    try:
        nodes = os.listdir(top)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(top, node, "properties")) if len(line.split()) >= 2)
        except OSError:      # not readable here: no count is better than a wrong one (a false refusal would waste a multi-GPU lease)
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no RANK / WORLD_SIZE in the environment): this process is the PARENT.
    It touches no GPU -- it has imported neither torch nor the prover library at this point -- and starts the ranks as a CHILD process,
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`
    (one process per GPU over RCCL; never an exec: a process that has initialised the GPU must not be replaced, and this one stays clean
    anyway), relays rank 0's single JSON line on stdout, everything else on stderr, and exits with the launcher's code.  Under an external
    launcher (the driver's torch.distributed.run) the ranks find RANK / WORLD_SIZE and this function is never reached."""
    import socket
    import subprocess
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--device", type=int, default=None)
    ap.add_argument("--mode", default="replicas")
    known, _ = ap.parse_known_args(argv)
    if known.gpus <= 1 or known.mode == "sharded-replay" or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return None
    if known.device is None:
        have = visible_gpus()
        if have is not None and have < known.gpus:
            print("bench.py: --gpus %d but only %d GPU(s) visible (KFD topology / *_VISIBLE_DEVICES); pass --device D to run all ranks on one "
                  "device on purpose" % (known.gpus, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(known.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), VPBS_BENCH_SELF_LAUNCHED="1")
    proc =subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited 0 without a JSON line", file=sys.stderr)
        rc = 1
    return rc


if __name__ == "__main__":
    _rc = launch_ranks(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if _cpu_share_before_hip() < 12 else "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: its bundled HIP runtime must be the one the prover library binds to)
import torch.distributed as dist  # noqa: E402

import vpbs_amd  # noqa: E402
from vpbs_amd import synth  # noqa: E402

def shares_the_gpu(ctx):
    """a context that proves beside others on the same GPU: their work hides its latency-bound phases, so it trades the 16-lane Poseidon form
    (a third of the latency, 3.7 x the instructions) for the one-lane form down to 2048 nodes (VPBS_OPT_WIDE_THRESHOLD; 8.33 -> 8.14-8.22 ms
    per chained proof with six chains, tools/experiments/wide_ab.sh).  An explicit VPBS_WIDE_THRESHOLD wins."""
    if "VPBS_WIDE_THRESHOLD" not in os.environ:
        ctx.set_option("wide_threshold", 2048)
    return ctx


STEPS_PER_VPBS = 730       # n + 2 with n = 728 (reference src/main.rs:27, ivc_based_vpbs.rs:433-436)
# Degree of the step circuit at N = 1024.  /root/reference/src/vtfhe/ivc_based_vpbs.rs:54-61 pads the common-data circuit with NoopGates
# until it HAS 2^15 gates and only then calls build(), which appends the public-input hash rows, the PublicInputGate and the constant
# gates and pads to the next power of two: 2^16 rows (plonky2's cyclic-recursion test uses the same idiom).  Independent check: the
# step logic without the recursive verifier, described gate by gate in circuitgen/step_circuit.py, already needs 38 312 rows at the paper's
# parameters (tests/test_gpu_step_circuit.py proves that circuit).  SURVEY.md 8d quotes 2^15; that size is kept as a secondary figure
# (`survey_degree_2pow15` in the JSON line, `--log-n 15`).
LOG_N = 16
SURVEY_LOG_N = 15
# the gate set of a recursive plonky2 circuit under standard_recursion_config (ivc_based_vpbs.rs:80-157 builds the step circuit from
# arithmetic / base-sum / Poseidon gadgets and a recursive verifier); selector_polynomials gives it 4 selector columns
GATES = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing",
         "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"]
N_CONSTANTS, N_ROUTED = 6, 80   # constants_sigmas = 4 selectors + 2 gate constants + 80 sigma columns
# public inputs of a step proof at N = 1024, K = 2 (ivc_based_vpbs.rs:196-207): acc_init 2048 + counter + current
# accumulator 2048 + two chain hashes 8 + verifier data 68
N_PUBLIC_INPUTS = 4173
COLS = dict(synth.STEP_COLS, constants_sigmas=N_CONSTANTS + N_ROUTED)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU ceilings, measured on gfx950 with tools/microbench_valu2.hip (profiles/r02_microbench_valu2.txt: distinct source registers per
# chain, in-kernel s_memtime / s_memrealtime, exact waves per SIMD, per-SIMD spans):
#  * v_fma_f32, v_add_u32, v_xor_b32: 2.24-2.27 cycles per wave64 instruction at 8 waves per SIMD -- the SIMD-32 rate of
#    MI355X_MICROARCH.md (256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s).  (Round 1's harness read one VGPR for two operands
#    and reported 4.13 for v_fma_f32: wrong.)
#  * every instruction the Poseidon / NTT arithmetic is made of -- v_mad_u64_u32, v_mul_lo/hi_u32, carry adds (v_add_co / v_addc_co),
#    v_cndmask_b32_e64, 64-bit shifts, three-operand integer VOP3 -- runs at HALF that rate: 4.15-4.26 cycles at 8 waves per SIMD,
#    4.3-4.5 at 4 waves per SIMD (the occupancy of the leaf-hash kernel), 4.6-5.1 at 2, 5.5-6.75 for a lone wave.
# valu_frac prices the kernel against the fp32 rate at the nominal clock (what the guide calls the vector peak).  int_issue_frac prices it
# against what the SIMD can issue AT THE CLOCK THE KERNEL REALLY RAN AT: 97 % of the permutation's VALU instructions (v_mad_u64_u32 54 %,
# v_cndmask_b32_e64 14 %, carry adds / subtracts 27 %: tools/count_poseidon_isa.py) are of the half-rate class -- 16 lanes per cycle, 4
# cycles per wave64 instruction, of which the microbenchmark reaches 4.15 -- the other 3 % (v_mov) full rate, 2 cycles.  The clock is
# measured by one wave of every timed launch over its own lifetime (s_memtime / s_memrealtime: vpbs_timing_shader_clock): a loaded MI355X
# sustains ~2.25 GHz under this kernel, not the 2.4 GHz an idle probe or rocm-smi shows (round 2's first figures assumed 2.39 GHz and a
# 4.2-cycle ceiling, two errors that cancelled), and ~2.0 GHz for several milliseconds after a pause (tools/experiments/idle_gap.py).
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
LEAF_HASH_INSTR_PER_PERM = 13335  # dynamic VALU instructions per permutation (tools/count_poseidon_isa.py; rounds 2-4: 15260)
HALF_RATE_SHARE = 0.97            # of those, the share that issues at 4 cycles per wave64 instruction; the rest at 2
CEILING_CYCLES_PER_INSTR = HALF_RATE_SHARE * 4.0 + (1 - HALF_RATE_SHARE) * 2.0
SCLK_FALLBACK_HZ = 2.25e9         # only if the in-kernel measurement is unavailable


def leaf_hash_bytes_per_step(log_n=LOG_N):
    """Algorithmic bytes of the dominant kernel (Poseidon leaf hashing) per step: every LDE element read once,
    one 32-byte digest written per leaf; three launches (wires, Z/pp, quotient)."""
    lde = 1 << (log_n + 3)
    return sum(lde * (COLS[k] * 8 + 32) for k in ("wires", "zs_partial_products", "quotient"))


def leaf_hash_perms_per_step(log_n=LOG_N):
    lde = 1 << (log_n + 3)
    return sum(lde * ((COLS[k] + 7) // 8) for k in ("wires", "zs_partial_products", "quotient"))


_STEP_CIRCUIT = {}


def step_circuit():
    """the step circuit at the paper's parameters as DATA: the exported description (verifiable-fhe-paper_amd/circuit_file.py; written by
    __graft_entry__.build() with tools/export_step_circuit.py, the stand-in for the Rust circuit builder) loaded through the product
    package -- nothing of a circuit builder is imported here"""
    if "circ" not in _STEP_CIRCUIT:
        from vpbs_amd import circuit_file
        t0 = time.perf_counter()
        _STEP_CIRCUIT["circ"] = circuit_file.load(circuit_file.find_step_circuit(1024, 2, 4, 5, 728))
        _STEP_CIRCUIT["seconds"] = time.perf_counter() - t0
    return _STEP_CIRCUIT["circ"], _STEP_CIRCUIT["seconds"]


def step_circuit_pipeline(device, proofs=24, witness_threads=6, provers=3):
    """The reference's step circuit without its recursive verifier (build_step_circuit, ivc_based_vpbs.rs:80-155, described by
    circuitgen/step_circuit.py at the paper's parameters: 38 312 gate rows, degree 2^16, 4 105 public inputs) through the whole product
    pipeline: compiled witness generation on host threads into pinned buffers -> H2D -> step proof on the device, `provers` prover
    contexts in flight.  Reported next to the headline: it is a different circuit (6 gate types, real copy constraints, no recursion
    rows) and includes the host stage and the PCIe copy that `value` excludes."""
    import queue
    import threading
    witness_threads = int(os.environ.get("VPBS_PIPE_WITNESS", witness_threads))
    provers = int(os.environ.get("VPBS_PIPE_PROVERS", provers))
    from vpbs_amd import api
    N, K, ELL, LOGB, n_lwe = 1024, 2, 4, 5, 728
    b, t_build = step_circuit()
    t0 = time.perf_counter()
    sigma = b.circuit.sigma_values()
    targets = b.preset_pos     # acc_init, acc_in, the GGSW, counter, mask, the two chain hashes: the exporter's order (ivc_based_vpbs.rs:325-330)
    plan = b.circuit.witness_plan(targets)
    t_plan = time.perf_counter() - t0
    pi_pos = np.array(b.pi_pos)
    pi_cols, pi_rows = pi_pos[:, 0], pi_pos[:, 1]
    n_constants = b.constants.shape[0]
    cs_values = np.concatenate([b.constants, sigma])
    d_sigma = torch.from_numpy(sigma.view(np.int64)).cuda(device)
    digest = np.array([11, 22, 33, 44], np.uint64)
    ctxs = [shares_the_gpu(vpbs_amd.Context(device, log_n_max=16)) for _ in range(provers)]
    css = [c.commit_values(cs_values) for c in ctxs]
    n_buf = witness_threads + provers + 1
    bufs = [torch.empty((135, b.n), dtype=torch.int64).pin_memory() for _ in range(n_buf)]
    views = [t.numpy().view(np.uint64) for t in bufs]
    rng = np.random.default_rng(2024)
    base = rng.integers(0, synth.P, size=len(targets), dtype=np.uint64)
    i_counter = len(targets) - 10

    def values(i):
        v = base.copy()
        v[i_counter] = 2 + i % n_lwe          # a CMUX step; every proof has its own accumulator / mask
        v[:2 * K * N] = rng.integers(0, synth.P, size=2 * K * N, dtype=np.uint64)
        return v

    free_q, ready_q = queue.Queue(), queue.Queue()
    for i in range(n_buf):
        free_q.put(i)
    todo = queue.Queue()
    vals = [values(i) for i in range(proofs + provers)]
    wit_ms, errs = [], []

    def witness_worker():
        try:
            while True:
                i = todo.get()
                if i is None:
                    return
                k = free_q.get()
                t = time.perf_counter()
                plan.run(vals[i], threads=2, out=views[k])
                wit_ms.append(1e3 * (time.perf_counter() - t))
                ready_q.put((k, views[k][pi_cols, pi_rows]))
        except Exception as e:
            errs.append(e)
            ready_q.put(None)

    done = []

    def prover(j):
        try:
            while True:
                item = ready_q.get()
                if item is None:
                    ready_q.put(None)
                    return
                k, pis = item
                si = ctxs[j].make_step_inputs(b.log_n, views[k], None, None, css[j], digest, pis, sigmas=int(d_sigma.data_ptr()), n_routed=N_ROUTED,
                                              n_constants=n_constants, gates=b.gates)
                proof = ctxs[j].prove_step(si)
                free_q.put(k)
                done.append((proof, pis))
        except Exception as e:
            errs.append(e)

    def run(count):
        for i in range(count):
            todo.put(i)
        ws = [threading.Thread(target=witness_worker) for _ in range(witness_threads)]
        ps = [threading.Thread(target=prover, args=(j,)) for j in range(provers)]
        for t in ws + ps:
            t.start()
        for _ in ws:
            todo.put(None)
        for t in ws:
            t.join()
        ready_q.put(None)
        for t in ps:
            t.join()
        ready_q.get()
        if errs:
            raise errs[0]

    run(provers)                      # warm-up: pools, tables, first-touch of the pinned buffers
    done.clear()
    wit_ms.clear()
    t0 = time.perf_counter()
    run(proofs)
    elapsed = time.perf_counter() - t0
    proof, pis = done[-1]
    ok = api.verify_step(proof, css[0].cap(), [n_constants + N_ROUTED, 135, 20, 16], digest, pis, b.log_n, check_permutation=True,
                         n_constants=n_constants, n_routed=N_ROUTED, gates=b.gates)
    for c, cs in zip(ctxs, css):
        cs.free()
        c.close()
    plan.free()
    if not ok:
        raise RuntimeError("step-circuit proof did not verify")
    return {"circuit": "build_step_circuit (ivc_based_vpbs.rs:80-155) at N=1024, k=1, ELL=4, LOGB=5, n=728, no recursive verifier: "
                       "%d gate rows, degree 2^%d, %d public inputs" % (b.used_rows, b.log_n, len(b.pi_pos)),
            "step_proofs_per_s": proofs / elapsed, "ms_per_step_proof": 1e3 * elapsed / proofs, "proofs": proofs,
            "witness_ms_per_proof_one_thread_pair": sum(wit_ms) / max(1, len(wit_ms)), "witness_threads": witness_threads, "provers": provers,
            "includes": "compiled witness generation (vpbs_witness_plan_run, host), H2D of the 70.8 MB wire matrix from pinned memory, "
                        "the step proof; the last proof is verified by vpbs_verify_step",
            "setup_s": {"circuit_file_load": t_build, "sigma_and_witness_plan": t_plan}}


def step_circuit_device_pipeline(device, batch=64, batches=4, provers=4):
    batch = int(os.environ.get("VPBS_PIPE_BATCH", batch))
    provers = int(os.environ.get("VPBS_PIPE_PROVERS", provers))
    """The same circuit with the witnesses generated ON THE DEVICE (vpbs_witness_device_*): `batch` PartialWitnesses per run of the level
    schedule (two device objects on their own contexts, double-buffered), each instance gathered into a device wire matrix and proven
    by one of `provers` prover contexts.  Nothing but the PartialWitness values (20 490 field elements per step) crosses PCIe."""
    import queue
    import threading
    from vpbs_amd import api
    N, K, ELL, LOGB, n_lwe = 1024, 2, 4, 5, 728
    b, _ = step_circuit()
    sigma = b.circuit.sigma_values()
    targets = b.preset_pos
    plan = b.circuit.witness_plan(targets)
    pi_pos = b.pi_pos
    n_constants = b.constants.shape[0]
    cs_values = np.concatenate([b.constants, sigma])
    d_sigma = torch.from_numpy(sigma.view(np.int64)).cuda(device)
    digest = np.array([11, 22, 33, 44], np.uint64)
    wctx = [vpbs_amd.Context(device, log_n_max=16) for _ in range(2)]
    wdev = [api.WitnessDevice(c, plan, batch) for c in wctx]
    pctx = [shares_the_gpu(vpbs_amd.Context(device, log_n_max=16)) for _ in range(provers)]
    css = [c.commit_values(cs_values) for c in pctx]
    d_wires = [torch.zeros((135, b.n), dtype=torch.int64, device="cuda:%d" % device) for _ in range(provers)]
    rng = np.random.default_rng(4048)
    base = rng.integers(0, synth.P, size=len(targets), dtype=np.uint64)

    def values(batch_index):
        v = np.repeat(base[:, None], batch, axis=1)
        v[:2 * K * N] = rng.integers(0, synth.P, size=(2 * K * N, batch), dtype=np.uint64)          # accumulators of every instance
        v[len(targets) - 10] = 2 + (batch_index * batch + np.arange(batch)) % n_lwe                # counters: CMUX steps
        return np.ascontiguousarray(v)

    free_obj, ready, errs, done, wit_s, primed = queue.Queue(), queue.Queue(), [], [], [], []
    for k in range(2):
        free_obj.put(k)
    outstanding = [0, 0]
    lock = threading.Lock()

    def witness_thread(n_batches):
        try:
            for bi in range(n_batches):
                k = free_obj.get()
                t = time.perf_counter()
                wdev[k].run(values(bi))
                wit_s.append(time.perf_counter() - t)
                if bi == 0:
                    primed.append(time.perf_counter())       # the pipeline is primed: steady state from here
                with lock:
                    outstanding[k] = batch
                for i in range(batch):
                    ready.put((k, i))
        except Exception as e:
            errs.append(e)
        for _ in range(provers):
            ready.put(None)

    def prover(j):
        try:
            while True:
                item = ready.get()
                if item is None:
                    return
                k, i = item
                wdev[k].wires(i, d_wires[j].data_ptr())
                pis = wdev[k].read(i, pi_pos)
                with lock:
                    outstanding[k] -= 1
                    if outstanding[k] == 0:
                        free_obj.put(k)
                si = pctx[j].make_step_inputs(b.log_n, d_wires[j].data_ptr(), None, None, css[j], digest, pis, on_device=True,
                                              shapes=(135, 20, 16), sigmas=int(d_sigma.data_ptr()), n_routed=N_ROUTED, n_constants=n_constants,
                                              gates=b.gates)
                done.append((pctx[j].prove_step(si), pis))
        except Exception as e:
            errs.append(e)

    def run(n_batches):
        ts = [threading.Thread(target=witness_thread, args=(n_batches,))] + [threading.Thread(target=prover, args=(j,)) for j in range(provers)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]

    run(1)                      # warm-up (graph capture, pools)
    done.clear()
    wit_s.clear()
    primed.clear()
    run(batches)
    elapsed = time.perf_counter() - primed[0]   # the first witness batch of a chain overlaps with whatever ran before it
    proof, pis = done[-1]
    ok = api.verify_step(proof, css[0].cap(), [n_constants + N_ROUTED, 135, 20, 16], digest, pis, b.log_n, check_permutation=True,
                         n_constants=n_constants, n_routed=N_ROUTED, gates=b.gates)
    for w in wdev:
        w.free()
    for c, cs in zip(pctx, css):
        cs.free()
        c.close()
    for c in wctx:
        c.close()
    plan.free()
    if not ok:
        raise RuntimeError("step-circuit proof (device witness) did not verify")
    proofs = batch * batches
    return {"step_proofs_per_s": proofs / elapsed, "ms_per_step_proof": 1e3 * elapsed / proofs, "proofs": proofs, "witness_batch": batch,
            "device_witness_ms_per_batch": 1e3 * sum(wit_s) / max(1, len(wit_s)), "provers": provers,
            "includes": "witness generation on the device for %d steps at a time (level schedule in a hipGraph, overlapped with the proofs of "
                        "the previous batch; timed from the moment the first batch is ready), gather of each instance's wires in HBM, the "
                        "step proof; the last proof is verified" % batch}


def batch_of_128(device, pools=(1, 2, 4, 8), proofs=128):
    """BASELINE config 3: a batch of 128 independent step proofs of ONE circuit (the N = 1024 shape: degree 2^16, 135 wire columns, 14 gate
    types; the constants/sigmas commitment is the circuit's and shared) on one GPU.  Every instance has its own seeded wire matrix
    (resident in HBM before the clock starts: 128 x 70.8 MB = 9.1 GB) and its own 4173 public inputs; a bounded pool of prover contexts
    (one HIP stream + host thread each, ~1.1 GB of HBM per context: LDEs 0.72 GB, coefficients, digests, quotient scratch) works through
    the queue.  Reported per pool size so that the saturation point is visible."""
    import queue
    import threading
    gates = vpbs_amd.api.GateSet(GATES)
    digest = np.array([11, 22, 33, 44], np.uint64)
    n = 1 << LOG_N
    cs_values = synth.step_inputs(LOG_N, cols=COLS)["constants_sigmas"]
    d_cs = torch.from_numpy(cs_values.view(np.int64)).cuda(device)
    sig_ptr = d_cs.data_ptr() + 8 * N_CONSTANTS * n
    gen = torch.Generator(device="cuda:%d" % device)
    wires, pis = [], []
    for i in range(proofs):   # seeded per instance, generated on the device (uniform below 2^63 < p: the cost of a proof does not depend on the data)
        gen.manual_seed(0x5EED0000 + 16 * i)
        wires.append(torch.randint(0, synth.P >> 1, (COLS["wires"], n), dtype=torch.int64, device="cuda:%d" % device, generator=gen))
        pis.append(synth.field_elements(0xABCD + i, N_PUBLIC_INPUTS))
    torch.cuda.synchronize()
    out = {"proofs": proofs, "hbm_resident_inputs_gb": proofs * COLS["wires"] * n * 8 / 1e9, "pools": {}}
    ctxs, css = [], []
    distinct = set()
    for pool in pools:
        while len(ctxs) < pool:
            c = shares_the_gpu(vpbs_amd.Context(device, log_n_max=16))
            c.set_gate_lanes(1)
            ctxs.append(c)
            css.append(c.commit_values(cs_values))
        todo = queue.Queue()
        for i in range(proofs):
            todo.put(i)
        errs, caps = [], [None] * proofs

        def work(k):
            try:
                while True:
                    try:
                        i = todo.get_nowait()
                    except queue.Empty:
                        return
                    si = ctxs[k].make_step_inputs(LOG_N, wires[i].data_ptr(), None, None, css[k], digest, pis[i], on_device=True,
                                                  shapes=(COLS["wires"], COLS["zs_partial_products"], COLS["quotient"]), sigmas=sig_ptr,
                                                  n_routed=N_ROUTED, n_constants=N_CONSTANTS, gates=gates)
                    caps[i] = ctxs[k].prove_step(si)["caps"][0, 0].tobytes()
            except Exception as e:
                errs.append(e)
        for c in ctxs[:pool]:
            c.synchronize()
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(k,)) for k in range(pool)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for c in ctxs[:pool]:
            c.synchronize()
        dt = time.perf_counter() - t0
        if errs:
            raise errs[0]
        distinct = set(caps)
        out["pools"][str(pool)] = {"seconds": dt, "step_proofs_per_s": proofs / dt, "ms_per_step_proof": 1e3 * dt / proofs,
                                   "vpbs_proofs_per_s": proofs / dt / STEPS_PER_VPBS}
    if len(distinct) != proofs:
        raise RuntimeError("batch of %d: only %d distinct wires caps" % (proofs, len(distinct)))
    best = max(out["pools"], key=lambda k: out["pools"][k]["step_proofs_per_s"])
    out["best_pool"] = int(best)
    out["step_proofs_per_s"] = out["pools"][best]["step_proofs_per_s"]
    out["hbm_per_context_gb"] = 1.1
    for cs, c in zip(css, ctxs):
        cs.free()
        c.close()
    del wires
    torch.cuda.empty_cache()
    return out


def reference_probe():
    """is the reference's own toolchain on this box?  `cargo --version` and the pinned crates (plonky2 0.2.0,
    /root/reference/Cargo.lock:371-374) in an offline registry.  The reference itself is never at /root/reference on the GPU box, and its
    `cargo run --release` is one whole vPBS at N = 1024 (the paper: ~20 min on 192 cores) -- outside any bench budget -- so even with a
    toolchain the figure reported next to the GPU's is the restated CPU prover; the probe records what was found."""
    import shutil
    import subprocess
    cargo = shutil.which("cargo")
    out = {"cargo": None, "plonky2_0_2_0_in_offline_registry": False, "reference_run": False}
    if cargo:
        try:
            out["cargo"] = subprocess.run([cargo, "--version"], capture_output=True, text=True, timeout=20).stdout.strip()
        except Exception as e:   # noqa: BLE001
            out["cargo"] = "present, but `cargo --version` failed: %s" % e
        reg = os.path.expanduser("~/.cargo/registry/src")
        if os.path.isdir(reg):
            out["plonky2_0_2_0_in_offline_registry"] = any(os.path.isdir(os.path.join(reg, d, "plonky2-0.2.0")) for d in os.listdir(reg))
    out["why_not_run"] = ("no Rust toolchain on this box" if not cargo else
                          "toolchain present; the reference's only binary proves one whole vPBS (730 steps, minutes to hours of CPU) and its "
                          "sources are not part of this repository's snapshot")
    return out


def cpu_baseline(gpu_proof=None, runs=5):
    """Complete step proofs on the host cores with the CPU oracle (kind 'port': the restated algorithm in C, OpenMP over every CPU the
    container may use, the Poseidon permutation eight at a time on AVX-512 lanes where the CPU has them -- oracle/poseidon_x8.c), timed
    stage by stage the way the reference prints its TimingTree (ivc_based_vpbs.rs:301,309,332,340); `runs` repetitions, the MEDIAN is
    reported.  gpu_proof = (proof, bytes, constants/sigmas cap) of the GPU for the same instance: compared word for word with the first
    run's proof (the bench fails if they differ)."""
    import statistics
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen")]   # the checker and the big-integer model it leans on
    import gates_oracle
    import oracle as orc
    import step_oracle
    orc.build()
    inputs = synth.step_inputs(LOG_N, cols=COLS)
    pis = synth.field_elements(0xABCD, N_PUBLIC_INPUTS)
    digest = np.array([11, 22, 33, 44], np.uint64)
    cs = orc.Batch(inputs["constants_sigmas"], 3, 4, True)   # committed once per circuit: untimed
    sig = np.ascontiguousarray(inputs["constants_sigmas"][N_CONSTANTS:N_CONSTANTS + N_ROUTED])
    gs = gates_oracle.GateSet(GATES)
    x8 = bool(orc.lib().orc_poseidon_x8_available())

    def one():
        """plonk/prover.rs prove() after witness generation, stage by stage (the transcript order of step_oracle.prove_step)"""
        T = {}

        def timed(name, f):
            t = time.perf_counter()
            r = f()
            T[name] = T.get(name, 0.0) + 1e3 * (time.perf_counter() - t)
            return r
        t_all = time.perf_counter()
        pi_hash = timed("public_inputs_hash", lambda: orc.hash_no_pad(pis))
        w = timed("wires_commit (ifft + lde + transpose + merkle)", lambda: orc.Batch(inputs["wires"], 3, 4, True))
        ch = orc.ChallengerState()
        ch.observe(digest); ch.observe(pi_hash); ch.observe(w.cap())
        betas, gammas = ch.get_n(2), ch.get_n(2)
        zs = timed("partial_products", lambda: orc.partial_products(inputs["wires"][:N_ROUTED], sig, betas, gammas))
        zb = timed("zs_partial_products_commit", lambda: orc.Batch(zs, 3, 4, True))
        ch.observe(zb.cap())
        alphas = ch.get_n(2)
        gt = timed("quotient: gate constraints on the coset", lambda: gs.terms_coset(cs.coeffs()[:N_CONSTANTS], w.coeffs(), pi_hash, alphas))
        q = timed("quotient: permutation terms, / Z_H, coset ifft", lambda: orc.quotient_permutation(
            w.coeffs()[:N_ROUTED], cs.coeffs()[N_CONSTANTS:N_CONSTANTS + N_ROUTED], zb.coeffs(), betas, gammas, alphas, gate_terms=gt))
        qb = timed("quotient_commit", lambda: orc.Batch(q, 3, 4, False))
        ch.observe(qb.cap())
        zeta = ch.get_ext()
        oracles = [cs, w, zb, qb]
        ncols = [o.ncols for o in oracles]
        batches, zeta_next = step_oracle.step_batches(ncols, 2, zeta, LOG_N)
        openings = timed("openings", lambda: np.concatenate([o.eval_ext(zeta) for o in oracles] + [zb.eval_ext(zeta_next)[:2]]))
        ch.observe(openings)
        fri = timed("fri: prove_openings (combine, folds, trees, PoW, queries)", lambda: orc.prove_openings(oracles, batches, ch, orc.fri_params(LOG_N), LOG_N))
        total = 1e3 * (time.perf_counter() - t_all)
        proof = {"caps": np.stack([w.cap(), zb.cap(), qb.cap()]), "openings": openings, "fri": fri,
                 "challenges": np.array(betas + gammas + alphas + [int(zeta[0]), int(zeta[1])], np.uint64), "cs_cap": cs.cap(), "ncols": ncols}
        return total, T, proof

    totals, stages, want = [], [], None
    for r in range(max(1, runs)):
        total, T, proof = one()
        totals.append(total)
        stages.append(T)
        if want is None:
            want = proof
    parity = None
    if gpu_proof is not None:
        # the oracle proved the very instance chain 0 of the GPU proved (same seeded columns, gates, public inputs, digest): every
        # word of the proof must agree -- caps, Fiat-Shamir challenges, openings, FRI proof (PoW nonce, query paths), serialised bytes
        got, got_bytes, got_cs_cap = gpu_proof
        bad = [k for k in ("caps", "challenges", "openings", "fri") if not (np.asarray(got[k]).reshape(-1) == np.asarray(want[k]).reshape(-1)).all()]
        if not (np.asarray(got_cs_cap) == want["cs_cap"]).all():
            bad.append("cs_cap")
        if got_bytes != step_oracle.to_bytes(want, want["ncols"], N_CONSTANTS, pis, LOG_N):
            bad.append("bytes")
        if bad:
            raise RuntimeError("full-size GPU step proof differs from the CPU oracle's in: " + ", ".join(bad))
        parity = {"compared": ["cs_cap", "caps", "challenges", "openings", "fri", "bytes"], "proof_bytes": len(got_bytes),
                  "fri_words": int(np.asarray(want["fri"]).size)}
    med = statistics.median(totals)
    return parity, {"value": (1e3 / med) / STEPS_PER_VPBS, "unit": "vPBS proofs/s", "ms_per_step": med,
            "cores": orc.effective_cpus(), "kind": "port", "runs": len(totals), "ms_per_step_runs": [round(t, 1) for t in totals],
            "stages_ms_median": {k: round(statistics.median(s[k] for s in stages), 2) for k in stages[0]},
            "poseidon": "oracle/poseidon_x8.c: eight permutations per AVX-512 register" if x8 else "oracle/poseidon.c: scalar (no AVX-512 on this CPU)",
            "sample": "%d complete step proofs of the SYNTHETIC step (2^%d rows, 135/20/16 columns, same seeded inputs as step_micro, partial "
                      "products, gate constraints of 14 gate types and quotient included; witness generation excluded, as in step_micro), median; C "
                      "oracle, OpenMP on every CPU the container may use (cgroup CPU quota; os.cpu_count() = %d)" % (len(totals), LOG_N, os.cpu_count() or 0),
            "reference_probe": reference_probe(),
            "note": "the restated CPU prover (not plonky2, not tuned beyond its Poseidon): FFTs are plain radix-2, the gate constraints are "
                    "evaluated in the extension-field form the verifier uses.  A stated baseline measured in the same run on the same box; the "
                    "optimisation target is the roofline fraction, and no GPU / CPU ratio is quoted"}


# Algorithmic bytes of ONE step proof at degree 2^16 by kernel group (MB; each datum a stage must touch counted once per stage: DESIGN.md
# "Kernels, bounds, algorithmic bytes"): what `roofline.step_hbm_frac` prices the WHOLE step with
STEP_ALGORITHMIC_MB = {"leaf hashing (3 launches)": 767.6, "Merkle levels": 100.0, "coset LDE of 171 columns": 806.0, "iNTT of 155 columns": 162.0,
                       "partial products": 220.0, "gate constraints": 591.0, "permutation quotient + combine + iNTT": 755.0,
                       "FRI combine / divide / fold / LDE / openings / queries": 330.0, "FRI round trees": 26.0}


def valu_budget(sclk_hz, step_ms, shared_gpu=False):
    """The WHOLE step against the bound that holds it: wave-level VALU instructions per step proof by kernel, from
    the committed SQ_INSTS_VALU pass over the synthetic step (profiles/rNN_pmc_sq_kernels.csv: rocprofv3 --pmc, tools/pmc_kernels.sh; a
    constant read from profiles/, like `traffic`), their sum / what 1024 SIMDs can issue at the clock measured inside this run's kernels =
    the time the step's instruction stream needs when nothing waits; `frac` = that time / the measured wall time per step proof.  With
    several chains per GPU every wait of one chain is filled by another, so frac is near 1 and throughput = instruction count."""
    import csv
    # shared_gpu: the counters of the settings a context runs with beside other chains (16-lane Poseidon threshold 2048, PoW in rounds:
    # tools/pmc_kernels.sh's third pass) -- the eight-chain headline's own instruction stream; else the defaults of a context alone on the GPU
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_sources
    paths = [kernel_sources.newest_profile("pmc_sq_kernels_shared_gpu.csv") if shared_gpu else None, kernel_sources.newest_profile("pmc_sq_kernels.csv")]
    # the counters are constants read from profiles/: do they belong to the device code of this checkout?  (tools/refresh_profiles.sh records the
    # fingerprint of the kernel sources next to them; the CPU suite fails on a mismatch, the line carries the answer)
    src = kernel_sources.newest_profile("pmc_sources.json")
    fresh = bool(src) and json.load(open(src)).get("sha256") == kernel_sources.fingerprint()["sha256"]
    for path in paths:
        if not path or not os.path.exists(path):
            continue
        name = os.path.basename(path)
        rows = list(csv.DictReader(open(path)))
        by = {r["kernel"]: r for r in rows}
        if "leaf_hash_kernel" not in by:
            continue
        if "valu_per_step_proof" in rows[0]:   # exact: whole periods of the step's kernel sequence, setup excluded (tools/pmc_table.py)
            per = {k: float(r["valu_per_step_proof"]) / 1e9 for k, r in by.items() if float(r["valu_per_step_proof"]) > 0}
            basis = "SQ_INSTS_VALU of the dispatches between the first and the last quotient_perm_kernel dispatch / the periods between them"
        else:                                   # tables of earlier rounds: all launches (the setup commitment's too) / (leaf-hash launches / 3)
            steps = float(by["leaf_hash_kernel"]["launches"]) / 3.0
            per = {k: float(r["SQ_INSTS_VALU"]) / steps / 1e9 for k, r in by.items() if float(r["SQ_INSTS_VALU"]) > 0}
            basis = "SQ_INSTS_VALU per kernel over all launches / (leaf-hash launches / 3)"
        total = sum(per.values())
        top = dict(sorted(per.items(), key=lambda kv: -kv[1])[:8])
        top["(all others)"] = total - sum(top.values())
        issue = 256 * 4 / CEILING_CYCLES_PER_INSTR * sclk_hz                # wave64 instructions per second, chip-wide
        floor_ms = total * 1e9 / issue * 1e3
        return {"bound": "int-valu-issue (whole step)", "wave_instructions_per_step_G": total, "by_kernel_G": top,
                "issue_peak_G_wave_instr_per_s": issue / 1e9, "shader_clock_mhz": sclk_hz / 1e6,
                "instruction_time_ms_per_step": floor_ms, "measured_ms_per_step_proof": step_ms, "frac": floor_ms / step_ms,
                "counters_match_kernel_sources": fresh,
                "counters_from": "profiles/" + name + " (" + basis + ", synthetic step; same kernels and "
                                 "shapes as the chained step; a constant read from profiles/, not re-measured in this run)",
                "what": "sum of the step's wave-level VALU instructions / (1024 SIMDs / %.2f cycles per instruction at the clock one wave of "
                        "every timed leaf-hash launch measured) / wall time per step proof of the timed region" % CEILING_CYCLES_PER_INSTR}
    return None


def roofline_of(per_step_ms, bytes_step, perms, sclk_mhz, sclk_samples, launches, measured_in, step_ms=None, shared_gpu=False):
    """The dominant kernel (Poseidon leaf hashing), priced as the contract asks -- ALGORITHMIC bytes per launch / average launch duration against
    the HBM peak -- with the bound that really holds it (integer VALU issue) beside it as `int_valu_issue`.
    per_step_ms: HIP-event time of its three launches per step proof; bytes_step / perms: algorithmic bytes and permutations per step;
    step_ms: wall time per step proof of the same region (for step_hbm_frac: the whole step against HBM)."""
    if per_step_ms <= 0:
        return {"bound": "hbm", "kernel": "leaf_hash_kernel", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}
    sclk_hz = sclk_mhz * 1e6 if sclk_mhz > 0 else SCLK_FALLBACK_HZ
    secs = per_step_ms * 1e-3
    lane_ops = perms * LEAF_HASH_INSTR_PER_PERM                      # one lane executes one permutation
    achieved = lane_ops / secs / 1e12
    # what 1024 SIMDs can issue for THIS instruction mix at the clock the kernel ran at: 64 lanes per wave64 instruction every
    # CEILING_CYCLES_PER_INSTR cycles (97 % of the stream is of the half-rate integer class: 16 lanes per cycle)
    peak = 256 * 4 * 64 / CEILING_CYCLES_PER_INSTR * sclk_hz / 1e12
    hbm = bytes_step / secs / 1e9
    traffic, traffic_from = None, None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_sources
    tpath = kernel_sources.newest_profile("pmc_leaf_hash.json")
    if tpath:
        # the counters were taken on whole launches at degree 2^16 (2^19 leaves); a rank of the coset-sharded step hashes its share of the
        # leaves per launch: the constant is scaled to the launch this run timed (it IS proportional: traffic = 1.00 x algorithmic bytes)
        traffic = json.load(open(tpath)).get("hbm_bytes_per_launch_avg")
        if traffic is not None:
            traffic *= bytes_step / float(leaf_hash_bytes_per_step(LOG_N))
        traffic_from = "profiles/" + os.path.basename(tpath)
    step_total = sum(STEP_ALGORITHMIC_MB.values()) * 1e6
    out = {"bound": "hbm", "kernel": "leaf_hash_kernel (Poseidon sponge over LDE rows, 3 launches per step proof)",
           "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS, "traffic": traffic,
           "algorithmic_bytes_per_launch": bytes_step / 3.0, "algorithmic_bytes_per_step": bytes_step,
           "launch_ms_avg": per_step_ms / 3.0, "kernel_ms_per_step": per_step_ms, "launches": launches, "measured_in": measured_in,
           "traffic_measured_in": (traffic_from or "not available") + ": rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; the guide's unit and gfx950 "
                                  "corrections) of this kernel at the same shapes in the synthetic step workload (bench.py --workload step under "
                                  "tools/pmc_kernels.sh), per launch on average; a constant read from profiles/, NOT re-measured inside this run",
           "why_so_low": "the kernel moves exactly its algorithmic bytes (traffic = 1.00 x) and is not a bandwidth kernel: ~%d VALU instructions per "
                         "64 absorbed bytes; the bound that holds it is integer instruction issue (int_valu_issue)" % LEAF_HASH_INSTR_PER_PERM,
           "int_valu_issue": {
               "bound": "int-valu-issue", "achieved": achieved, "peak": peak, "unit": "T lane-ops/s", "frac": achieved / peak,
               "note": "~%d VALU instructions per permutation, 97 %% of them (v_mad_u64_u32, carry adds, v_cndmask, VOP3 integer) issuing at 4 "
                       "cycles per wave64 instruction on gfx950 (profiles/r02_microbench_valu2.txt), the rest at 2.  peak = 1024 SIMDs x 64 lanes / "
                       "%.2f cycles at the shader clock one wave of every timed launch measured over its own lifetime.  An instruction-count "
                       "fraction says the kernel saturates the issue ports, not that the instruction stream is minimal; valu_frac prices the "
                       "same stream against the guide's fp32 vector peak (256 CU x 4 SIMD x 32 lanes x 2.4 GHz), which this instruction class "
                       "cannot reach" % (LEAF_HASH_INSTR_PER_PERM, CEILING_CYCLES_PER_INSTR),
               "poseidon_permutations_per_s": perms / secs,
               "valu_peak_tlaneops": VALU_PEAK_TLANEOPS, "valu_frac": achieved / VALU_PEAK_TLANEOPS,
               "shader_clock_mhz_in_kernel": sclk_mhz, "shader_clock_samples": sclk_samples,
               "cycles_per_valu_instr_per_simd": (secs * sclk_hz * 256 * 4) / (lane_ops / 64)},
           # kept at the top level for the scripts under tools/ that read them
           "int_issue_frac": achieved / peak, "shader_clock_mhz_in_kernel": sclk_mhz}
    if step_ms:
        budget = valu_budget(sclk_hz, step_ms, shared_gpu)
        if budget is not None:
            out["valu_budget"] = budget
        out["step_hbm_frac"] = step_total / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["step_algorithmic_bytes"] = {"total": step_total, "by_group_MB": STEP_ALGORITHMIC_MB,
                                         "what": "algorithmic bytes of one whole step proof / wall time per step proof of the timed region / 8 TB/s"}
    return out


def host_info():
    return {"hardware_threads": os.cpu_count(), "cgroup_cpu_max": (open("/sys/fs/cgroup/cpu.max").read().strip()
                                                                   if os.path.exists("/sys/fs/cgroup/cpu.max") else None),
            "loadavg_1min": os.getloadavg()[0]}


IVC_N, IVC_K, IVC_ELL, IVC_LOGB, IVC_NLWE = 1024, 2, 4, 5, 728   # the paper's parameters (/root/reference/src/main.rs:23-30)


def measure_ivc(args, rank, local_rank, world, distributed):
    """The headline: the reference's own object -- vPBS proofs as IVC chains (verified_pbs, ivc_based_vpbs.rs:159-386) through the library's
    driver vpbs_ivc_prove_pbs.  A step = one CHAINED step proof of the cyclic circuit (step logic + in-circuit verifier of the previous
    proof, 46 656 gate rows, degree 2^16) of every chain on this GPU: witness generation (host, two phases), upload, proof -- everything a
    step of the chain costs, the chain dependency included.  --warmup chained steps run untimed (after the base proof), then exactly --steps
    are timed between barrier + synchronise on both sides; the clock is placed from the library's progress hook."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prove_ivc
    from vpbs_amd import api, circuit_file
    N, K, ELL, LOGB, n_lwe, log_n = IVC_N, IVC_K, IVC_ELL, IVC_LOGB, IVC_NLWE, LOG_N
    total = n_lwe + 2
    W, Kt = args.warmup, args.steps
    # The chain is a pipeline: the early witness phase of a step runs AHEAD of its proof -- up to three steps ahead on the host (the three wire
    # matrices of vpbs_ivc_prove_pbs), up to two BATCHES ahead on the device (vpbs_ivc_set_device_witness: two device objects).  A timed window
    # is only fair in steady state: as much early-phase work for LATER steps falls into it as was done for ITS steps before it started.  So
    # (ADVICE r03): `run_in` untimed steps bring the pipeline to steady state before the --warmup steps (with the device pipeline the first
    # two batches start at once, ahead of everything: they must have been consumed), and the chain goes on for `tail` untimed steps after
    # the window, so that the early phases of later steps keep running inside it exactly as they would in a whole chain.
    ahead = 2 * args.device_witness if args.device_witness else 3
    run_in, tail = (ahead if args.device_witness else 0), ahead
    if run_in + W + Kt + tail > total:
        raise SystemExit("bench.py: run-in %d + --warmup %d + --steps %d + tail %d exceeds the %d steps of one vPBS chain at the paper's parameters"
                         % (run_in, W, Kt, tail, total))
    steps = run_in + W + Kt + tail
    sharded = distributed and args.mode == "sharded"
    n_chains = 1 if sharded else max(1, args.chains)
    cyc_path, dummy_path = circuit_file.find_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n)   # exported by __graft_entry__.build()
    t_setup = time.perf_counter()
    if distributed:
        # one process per GPU: every rank would size its witness pools for the whole machine -- each gets its share of the CPUs instead
        api.host_set_cpu_budget(max(2, api.host_set_cpu_budget(0) // world))
    # one chain per GPU with the CPUs to spare: 14 threads for the late witness phase (its last stage is 28 independent FRI queries); several
    # chains per GPU: the default of 8 (14 each cost six chains 4 % of their throughput: tools/experiments/chains6_late_ab.sh)
    api.host_set_late_threads(api.late_threads_for(n_chains, api.host_cpu_budget()))
    api.host_set_early_threads(api.early_threads_for(n_chains))
    chains, comm, native_comm = [], None, False
    for ci in range(n_chains):
        ctx = vpbs_amd.Context(local_rank, log_n_max=16)
        if n_chains > 1:
            ctx.set_gate_lanes(1)
        if sharded:
            from vpbs_amd import sharding
            stage_words = (2 << (log_n + 3)) // world
            native_comm = args.dist_backend == "nccl" and os.environ.get("VPBS_COMM", "rccl") == "rccl"
            comm = sharding.make_comm_rccl(ctx, stage_words=stage_words) if native_comm else \
                sharding.make_comm(device=torch.device("cuda", local_rank) if args.dist_backend == "nccl" else None, stage_words=stage_words,
                                   stage_device=torch.device("cuda", local_rank))
        cd, dd = circuit_file.load(cyc_path), circuit_file.load(dummy_path)
        if n_chains > 1:
            shares_the_gpu(ctx)
        ivc = api.Ivc(ctx, cd, dd, N, K, K * ELL * K * N, comm)
        if args.device_witness:
            ivc.set_device_witness(ELL, LOGB, args.device_witness, args.device_late)
        inst = 0 if sharded else rank * n_chains + ci        # every chain of every rank is its own PBS: own keys, own message
        keys = ctx.keygen(N, K, ELL, LOGB, n_lwe, 0x5EED0728 + inst, 4.99027217501041e-8, 1.17021618159313e-5)
        testv, delta = api.testv(N, 2)
        message = (1 + inst) % 2
        ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta * message % api.P)
        chains.append({"ctx": ctx, "ivc": ivc, "d": cd, "keys": keys, "testv": testv, "delta": delta, "ct": ct, "message": message})
    t_setup = time.perf_counter() - t_setup
    torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        for c in chains:
            c["ctx"].synchronize()

    gate = threading.Barrier(n_chains)
    clock = {}
    errs = []

    dominant = {"ms": 0.0, "count": 0}

    def on_step(ci, done):
        if done == run_in + W:
            # the warm-up steps are done on this chain: all chains of this process meet, the ranks meet, the device is drained -- t0
            gate.wait()
            if ci == 0:
                barrier()
                for c in chains:
                    c["ctx"].timing_enable(2)      # HIP events around the dominant kernel only, on each prover's own stream
                    c["ctx"].timing_report()
                clock["t0"] = time.perf_counter()
            gate.wait()
        elif done == run_in + W + Kt:
            # exactly --steps chained proofs of every chain later: they meet again, the ranks meet, the device is drained -- t1; the chains then
            # run on, untimed, for the tail
            gate.wait()
            if ci == 0:
                barrier()
                clock["t1"] = time.perf_counter()
                clock["sclk"] = chains[0]["ctx"].timing_shader_clock()
                for c in chains:
                    d = c["ctx"].timing_report().get("leaf_hash", {"ms": 0.0, "count": 0})
                    dominant["ms"] += d["ms"]; dominant["count"] += d["count"]
                    c["ctx"].timing_enable(0)
            gate.wait()

    def chain_thread(ci):
        try:
            torch.cuda.set_device(local_rank)
            c = chains[ci]
            c["ivc"].on_step(lambda done: on_step(ci, done))
            c["blob"], c["timing"] = c["ivc"].prove_pbs(c["testv"], c["ct"], c["keys"]["bsk"], c["keys"]["ksk"], steps)
            c["t_end"] = time.perf_counter()
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
            gate.abort()

    barrier()
    # every chain on a thread of its own; the interpreter's main thread only waits (a chain on the main thread used 0.85 of a CPU where the
    # others use 0.2: tools/prove_ivc.py VPBS_CPU_BY_ROLE at 2 CPUs)
    threads = [threading.Thread(target=chain_thread, args=(ci,)) for ci in range(n_chains)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    barrier()
    elapsed = clock["t1"] - clock["t0"]
    sclk_mhz, sclk_samples = clock["sclk"]
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # after the clock: the last proof of every chain is the proof of a prefix of a vPBS -- parsed, fully verified (gate constraints at
    # zeta included), its public inputs checked against the native accumulator chain and both native hash chains
    verify_ms = []
    for c in chains:
        vk, _ = c["ivc"].verifier_data()
        tv, _ = prove_ivc.check_chain(c["ctx"], c["d"], vk, c["blob"], c["keys"], c["testv"], c["delta"], c["ct"], N, n_lwe, log_n, steps, c["message"])
        verify_ms.append(1e3 * tv)
    result = None
    if rank == 0:
        d0 = chains[0]["d"]
        scale = 1.0 / world if sharded else 1.0     # sharded: rank 0 hashed 1/world of the leaves
        cols = {"wires": 135, "zs_partial_products": 20, "quotient": 16}
        lde = 1 << (log_n + 3)
        bytes_step = sum(lde * (cols[k] * 8 + 32) for k in cols) * scale
        perms = sum(lde * ((cols[k] + 7) // 8) for k in cols) * scale
        per_step_ms = dominant["ms"] / max(1, Kt * n_chains)
        steps_total = Kt * (1 if sharded else world) * n_chains
        t0s = chains[0]["timing"]
        result = {
            "metric": "vPBS proofs/sec at N=1024", "value": steps_total / elapsed / STEPS_PER_VPBS, "unit": "vPBS proofs/s",
            "n_gpus": world, "steps": Kt, "warmup": W, "ms_per_step": elapsed / Kt * 1e3,
            "ms_per_step_proof": elapsed / (Kt * n_chains) * 1e3,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks mod p)",
            "data": "synthetic", "step_proofs_per_s": steps_total / elapsed, "steps_per_vpbs_proof": STEPS_PER_VPBS,
            "config": {"workload": "N=1024 vPBS as the reference produces it (BASELINE config 2; verified_pbs, ivc_based_vpbs.rs:159-386): "
                                   "CHAINED step proofs of the cyclic step circuit (step logic + in-circuit verifier of the previous proof: "
                                   "%d gate rows, degree 2^%d, LDE 2^%d, 135 wire + 20 Z/partial-product + 16 quotient columns committed per "
                                   "step, %d public inputs) at k=1, ELL=4, LOGB=5, n=728, driven by vpbs_ivc_prove_pbs; %d chain(s) per GPU, "
                                   "each its own PBS (seeded keys, ciphertext); a step = one chained proof of every chain; value = "
                                   "chained step proofs/s / 730 (one vPBS = base proof + 730 chained proofs)"
                                   % (d0.meta.get("used_rows", 0), log_n, log_n + 3, len(d0.pi_pos), n_chains),
                       "stages": "per chained step, all inside the timed region: PartialWitness (previous proof's words + public inputs, GGSW, "
                                 "mask, verifier data) -> " + (
                                     "witness generation: the early phase of %d steps at a time ON THE DEVICE (two batches ahead of the chain; the "
                                     "chain's public inputs natively: accumulator chain on the device, hash chains on a host thread), each step's "
                                     "wires gathered on the device; the late phase = the in-circuit verifier's rows on the host, in stages as the "
                                     "previous proof's sections become final, packed values scattered on the device" % args.device_witness
                                     if args.device_witness else
                                     "witness generation on the host (early phase ahead on a second thread; late phase = the in-circuit verifier's rows, "
                                     "in stages as the previous proof's sections become final) -> wires to the device (early matrix in the "
                                     "background, the late values packed and scattered)") +
                                 " -> wires commit (iNTT + coset LDE + Poseidon Merkle) -> betas/gammas -> "
                                 "permutation Z + partial products -> commit -> alphas -> quotient polynomials (constraints of the "
                                 "circuit's %d gate types + permutation argument on the LDE coset, / Z_H, coset iNTT, 16 chunks) -> commit -> "
                                 "zeta -> openings -> FRI (combine, 3 arity-16 folds, 16-bit PoW, 28 queries), Fiat-Shamir transcript "
                                 "included.  Before the clock: key generation, circuit commitments, witness plans, the base proof, %d run-in "
                                 "steps (pipeline to steady state) and --warmup chained steps; after it the chain runs on for %d untimed steps, so "
                                 "that the early phases of later steps fall into the window as in a whole chain" % (d0.gates.n, run_in, tail),
                       "run_in_steps": run_in, "tail_steps": tail,
                       "parallelism": ("coset-sharded: ONE chain, every step proof split over %d GPUs (3 all-gathers of cap hashes, 1 device "
                                       "all-gather of quotient values, 1 all-reduce of query records per step; %s); every rank generates the "
                                       "identical witness" % (world, "native RCCL (dlopen) on the prover's stream" if native_comm else args.dist_backend))
                                      if sharded else "replicas: %d independent chain(s) per GPU, no data-path collective" % n_chains,
                       "chains_per_gpu": n_chains,
                       "early_witness_phase": ("on the device, %d steps per batch (vpbs_ivc_set_device_witness): %s"
                                               % (args.device_witness, "the late phase there as well" if args.device_late else "the host runs the late phase only"))
                                              if args.device_witness else "on the host (a second thread per chain)"},
            "roofline": roofline_of(per_step_ms, bytes_step, perms, sclk_mhz, sclk_samples, dominant["count"],
                                    "the timed chained steps of this run (HIP events on each prover's stream)",
                                    step_ms=elapsed / (Kt * n_chains) * 1e3, shared_gpu=n_chains > 1 and "VPBS_WIDE_THRESHOLD" not in os.environ),
            "chain_ms_per_step_split": {"witness_late_phase_host": t0s["late_witness_ms"], "late_rows_to_device": t0s["late_rows_upload_ms"],
                                        "prove_step": t0s["prove_step_ms"], "witness_early_phase_on_a_second_thread": t0s["early_witness_ms"],
                                        "base_proof_once": t0s["base_proof_ms"],
                                        "late_stages_run_during_the_previous_proofs_fri_stage": t0s["late_ahead_ms"],
                                        "over": "chain 0, warm-up steps included"},
            "chain_checks": {"last_proof_verify_ms": verify_ms, "proof_bytes": len(chains[0]["blob"]),
                             "what": "after the clock, per chain: the last proof (step %d of the chain) serialised by the library, parsed back, fully "
                                     "verified by vpbs_verify_step (gate constraints at zeta included); test vector, counter = %d, verifier data, the "
                                     "native accumulator chain and both native hash chains match its public inputs" % (steps, steps)},
            "before_the_clock_s": {"circuit_files_commit_plans_keys": t_setup},
            "host": host_info(),
        }
    for c in chains:
        c["ivc"].free()
    if native_comm and comm is not None:
        from vpbs_amd import sharding
        sharding.free_comm_rccl(comm)
    for c in chains:
        c["ctx"].close()
    chains.clear()
    torch.cuda.empty_cache()
    return result


def measure_step(args, rank, local_rank, world, distributed, log_n):
    """The synthetic step (rounds 1-2's headline, now `step_micro`): back-to-back step proofs of BASELINE config 2's shape with the wires
    resident in HBM -- no witness generation, no chain dependency: what the device-side prover does per step, at boost clock.  --workload
    step makes it the printed line (profiling scripts, the sharded-step measurements)."""
    sharded = distributed and args.mode == "sharded"
    comm = None
    native_comm = False
    if sharded:
        from vpbs_amd import sharding
    n_chains = 1 if sharded else max(1, args.step_chains)
    digest = np.array([11, 22, 33, 44], np.uint64)
    gates = vpbs_amd.api.GateSet(GATES)
    assert gates.num_selectors + gates.num_constants == N_CONSTANTS
    ctxs, sis, keep = [], [], []
    for c in range(n_chains):
        ctx = vpbs_amd.Context(local_rank, log_n_max=16)
        if n_chains > 1:
            ctx.set_gate_lanes(1)   # several chains keep the device busy by themselves: one stream per context is the better setting
        if sharded:
            # the collectives of the sharded step: the library's own RCCL path on GPUs (ncclAllGather / ncclAllReduce between device
            # buffers on the prover's stream; torch.distributed only carries the ncclUniqueId), host callbacks over torch.distributed for gloo
            stage_words = (2 << (log_n + 3)) // world
            native_comm = args.dist_backend == "nccl" and os.environ.get("VPBS_COMM", "rccl") == "rccl"
            if native_comm:
                comm = sharding.make_comm_rccl(ctx, stage_words=stage_words)
            else:
                comm = sharding.make_comm(device=torch.device("cuda", local_rank) if args.dist_backend == "nccl" else None,
                                          stage_words=stage_words, stage_device=torch.device("cuda", local_rank))
        # chain c of rank r proves its own seeded instance
        inst = 0 if sharded else rank * n_chains + c   # sharded: every rank works on the same proof
        inputs = synth.step_inputs(log_n, instance=inst, cols=COLS)
        dev = {k: torch.from_numpy(inputs[k].view(np.int64)).cuda() for k in ("wires", "quotient", "constants_sigmas")}
        if sharded:
            cs, _ = sharding.sharded_commit(ctx, dev["constants_sigmas"].data_ptr(), COLS["constants_sigmas"], log_n,
                                            device=torch.device("cuda", local_rank) if args.dist_backend == "nccl" else None)
        else:
            cs = ctx.commit_values(inputs["constants_sigmas"])      # once per circuit, untimed
        pis = synth.field_elements(0xABCD + inst, N_PUBLIC_INPUTS)
        sig_ptr = dev["constants_sigmas"].data_ptr() + 8 * N_CONSTANTS * (1 << log_n)   # sigma value columns
        si = ctx.make_step_inputs(log_n, dev["wires"].data_ptr(), None, None, cs, digest, pis,
                                  on_device=True, shapes=(COLS["wires"], COLS["zs_partial_products"], COLS["quotient"]),
                                  sigmas=sig_ptr, n_routed=N_ROUTED, n_constants=N_CONSTANTS, gates=gates)
        ctxs.append(ctx); sis.append(si); keep.append((dev, cs, pis))
    torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        for ctx in ctxs:
            ctx.synchronize()

    def run_steps(k):
        """k step proofs on every chain; chains run concurrently (ctypes releases the GIL inside the library)."""
        if n_chains == 1:
            for _ in range(k):
                ctxs[0].prove_step(sis[0], comm)
            return
        import threading
        errs = []

        def work(i):
            try:
                for _ in range(k):
                    ctxs[i].prove_step(sis[i])
            except Exception as e:  # surfaced below: a failed chain must fail the bench
                errs.append(e)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(n_chains)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        if errs:
            raise errs[0]

    run_steps(args.warmup)
    for ctx in ctxs:
        ctx.timing_enable(2)    # HIP events around the dominant kernel only, on each prover's own stream
        ctx.timing_report()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    dominant = {"ms": 0.0, "count": 0}
    sclk_mhz, sclk_samples = ctxs[0].timing_shader_clock()   # measured by one wave of every leaf-hash launch of the timed region
    for ctx in ctxs:
        d = ctx.timing_report().get("leaf_hash", {"ms": 0.0, "count": 0})
        dominant["ms"] += d["ms"]; dominant["count"] += d["count"]
        ctx.timing_enable(0)

    batch_result = None
    if world == 1 and n_chains == 1 and args.batch_chains > 1:
        extra = []
        for c in range(1, args.batch_chains):
            cx = shares_the_gpu(vpbs_amd.Context(local_rank, log_n_max=16))
            cx.set_gate_lanes(1)
            inp = synth.step_inputs(log_n, instance=c, cols=COLS)
            dv = {k: torch.from_numpy(inp[k].view(np.int64)).cuda() for k in ("wires", "quotient", "constants_sigmas")}
            csb = cx.commit_values(inp["constants_sigmas"])
            pi2 = synth.field_elements(0xABCD + c, N_PUBLIC_INPUTS)
            sp = dv["constants_sigmas"].data_ptr() + 8 * N_CONSTANTS * (1 << log_n)
            extra.append((cx, cx.make_step_inputs(log_n, dv["wires"].data_ptr(), None, None, csb, digest, pi2,
                                                  on_device=True, shapes=(COLS["wires"], COLS["zs_partial_products"], COLS["quotient"]),
                                                  sigmas=sp, n_routed=N_ROUTED, n_constants=N_CONSTANTS, gates=gates), dv, csb, pi2))
        ctxs += [e[0] for e in extra]; sis += [e[1] for e in extra]
        n_chains = len(ctxs)
        run_steps(1)
        barrier()
        tb = time.perf_counter()
        run_steps(args.steps)
        barrier()
        eb = time.perf_counter() - tb
        batch_result = {"chains": n_chains, "step_proofs_per_s": args.steps * n_chains / eb,
                        "vpbs_proofs_per_s": args.steps * n_chains / eb / STEPS_PER_VPBS,
                        "ms_per_step_proof": eb / (args.steps * n_chains) * 1e3}
        n_chains = 1

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel breakdown of one extra (untimed) step proof on chain 0 alone, for the record
    ctx = ctxs[0]
    ctx.timing_enable(1)
    proof0 = ctx.prove_step(sis[0], comm)
    breakdown = {k: round(v["ms"], 4) for k, v in ctx.timing_report().items()}
    gpu_proof = None
    if rank == 0 and not sharded and world == 1:   # kept for the word-for-word comparison with the oracle's proof (cpu_baseline leg)
        gpu_proof = (proof0, ctx.step_proof_to_bytes(sis[0], N_CONSTANTS, proof0), keep[0][1].cap().copy())
    ctx.timing_enable(0)

    out = None
    if rank == 0:
        steps_total = args.steps * (1 if sharded else world) * n_chains
        step_rate = steps_total / elapsed
        scale = 1.0
        per_step_ms = dominant["ms"] / max(1, args.steps * n_chains)  # three leaf_hash launches per step proof
        if sharded:
            scale /= world   # rank 0 hashed 1/world of the leaves
        out = {
            "metric": "vPBS proofs/sec at N=1024", "value": step_rate / STEPS_PER_VPBS, "unit": "vPBS proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_proof": elapsed / (args.steps * n_chains) * 1e3,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks mod p)",
            "data": "synthetic", "step_proofs_per_s": step_rate, "steps_per_vpbs_proof": STEPS_PER_VPBS,
            "config": {"workload": "SYNTHETIC step (not the chain): N=1024 vPBS step proof on 1xMI355X per rank, back to back on seeded random "
                                   "columns resident in HBM: degree 2^%d, LDE 2^%d, 135 wire + 20 Z/partial-product + 16 quotient columns "
                                   "committed per step, %d constant/sigma columns precommitted, 14 gate types, %d public inputs; value = "
                                   "step rate / 730; no witness generation, no chain dependency" % (log_n, log_n + 3, COLS["constants_sigmas"], N_PUBLIC_INPUTS),
                       "stages": "wires commit (iNTT + coset LDE + Poseidon Merkle) -> betas/gammas -> permutation Z + partial "
                                 "products (GPU) -> commit -> alphas -> quotient polynomials (GPU: constraints of %d gate types with "
                                 "their selector filters + the permutation argument over the LDE coset, / Z_H, coset iNTT, 16 "
                                 "chunks) -> commit -> zeta -> openings at zeta/g*zeta -> FRI (combine, 3 arity-16 folds, 16-bit "
                                 "PoW, 28 queries), Fiat-Shamir transcript included.  NOT in the timed region: witness generation" % gates.n,
                       "parallelism": ("coset-sharded: one chain, every commitment split over %d GPUs; per step 3 all-gathers of cap "
                                       "hashes, 1 device all-gather of quotient values (4 MiB) + 1 all-reduce of query records (%s)"
                                       % (world, "native RCCL (dlopen) on the prover's stream" if native_comm else args.dist_backend)) if sharded else
                                      "replicas: %d independent chain(s) per GPU, no data-path collective" % n_chains,
                       "chains_per_gpu": n_chains},
            "roofline": roofline_of(per_step_ms, leaf_hash_bytes_per_step(log_n) * scale, leaf_hash_perms_per_step(log_n) * scale, sclk_mhz,
                                    sclk_samples, dominant["count"], "the timed back-to-back synthetic steps"),
            "kernel_ms_one_step": breakdown,
            "host": host_info(),
        }
        if batch_result is not None:
            out["batch"] = batch_result   # several independent synthetic chains in flight (extra contexts / streams)
    if native_comm and comm is not None:
        sharding.free_comm_rccl(comm)
    state = {"ctxs": ctxs, "keep": keep, "gates": gates, "digest": digest, "gpu_proof": gpu_proof}
    return out, state


def survey_size_leg(local_rank, args, state):
    """the size SURVEY.md 8d quotes for N = 1024 (degree 2^15), same columns and gates, single chain: a secondary figure"""
    gates, digest, keep = state["gates"], state["digest"], state["keep"]
    c15 = vpbs_amd.Context(local_rank, log_n_max=16)
    i15 = synth.step_inputs(SURVEY_LOG_N, cols=COLS)
    d15 = {k: torch.from_numpy(i15[k].view(np.int64)).cuda() for k in ("wires", "constants_sigmas")}
    cs15 = c15.commit_values(i15["constants_sigmas"])
    si15 = c15.make_step_inputs(SURVEY_LOG_N, d15["wires"].data_ptr(), None, None, cs15, digest, keep[0][2], on_device=True,
                                shapes=(COLS["wires"], COLS["zs_partial_products"], COLS["quotient"]),
                                sigmas=d15["constants_sigmas"].data_ptr() + 8 * N_CONSTANTS * (1 << SURVEY_LOG_N),
                                n_routed=N_ROUTED, n_constants=N_CONSTANTS, gates=gates)
    for _ in range(max(1, args.warmup)):
        c15.prove_step(si15)
    torch.cuda.synchronize()
    t15 = time.perf_counter()
    for _ in range(args.steps):
        c15.prove_step(si15)
    c15.synchronize()
    e15 = (time.perf_counter() - t15) / args.steps
    cs15.free()
    c15.close()
    return {"ms_per_step_proof": e15 * 1e3, "step_proofs_per_s": 1.0 / e15, "vpbs_proofs_per_s": 1.0 / e15 / STEPS_PER_VPBS, "chains": 1}


def subprocess_json(cmd, env, timeout):
    import subprocess
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        return json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-500:]}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def launch_report(args, rank, local_rank, world, distributed):
    """What ran, for the reader of an N-GPU line (every rank calls this: it holds a collective): the ranks the communication library itself
    counted (a one-word all-reduce of ones over the process group -- RCCL for the default backend), its version, the devices, who started the
    ranks, the CPUs each rank was given.  The pipeline chosen from that share is added by main() once it is decided."""
    info = {"ranks": 1, "backend": None, "what": "one-word all-reduce of ones over the process group of this run"}
    if distributed:
        ones = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local_rank) if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        devs = [None] * world
        dist.all_gather_object(devs, "%s:%d" % (os.uname().nodename, local_rank))
        info.update(ranks=int(ones.item()), backend="rccl (torch.distributed 'nccl')" if args.dist_backend == "nccl" else args.dist_backend,
                    devices=devs)
    else:
        info["what"] = "no process group at N = 1"
    try:
        info["version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:   # noqa: BLE001
        info["version"] = "unavailable (%s)" % type(e).__name__
    return {"rccl": info,
            "launched_by": "bench.py itself (child process: python -m torch.distributed.run, one rank per GPU)" if os.environ.get("VPBS_BENCH_SELF_LAUNCHED")
                           else ("an external launcher (RANK / WORLD_SIZE in the environment)" if distributed else "a plain process")}


def sharded_ceiling(world):
    """--mode sharded is ONE chain split over the GPUs (strong scaling, a latency tool): what a rank still has to compute, measured by replay on
    one GPU with the collectives answered from a recording -- the ceiling of the speed-up, communication excluded"""
    for name in ("r05_sharded_rank_times.json", "r04_sharded_rank_times.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            d = json.load(open(path))
            w = d.get("worlds", {}).get(str(world))
            if not w:
                return {"from": "profiles/" + name, "error": "no replay for world %d" % world}
            return {"from": "profiles/" + name, "single_gpu_ms_per_step": d["single_gpu"]["wall_ms"], "slowest_rank_ms_per_step": w["T_rank_ms"],
                    "compute_only_speedup_ceiling": w["compute_speedup_vs_single_gpu"],
                    "replicated_kernel_ms": w["kernel_ms_replicated_groups"], "sharded_kernel_ms": w["kernel_ms_sharded_groups"],
                    "what": "per-rank compute + transcript round trips of the coset-sharded synthetic step, each rank alone on one MI355X, "
                            "collectives answered from a recording (bench.py --mode sharded-replay): the work that does not shrink with the "
                            "number of ranks (tree tops, FRI trees, iNTT, partial products, PoW) bounds the strong-scaling speed-up here; "
                            "near-linear scaling of vPBS proofs/s is what --mode replicas (the default) gives"}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60, help="timed steps.  Default workload: CHAINED step proofs of every IVC chain on the GPU")
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", choices=["ivc", "step"], default="ivc",
                    help="ivc (default, the headline): vPBS proofs as the reference's IVC chains through vpbs_ivc_prove_pbs -- witness generation, "
                         "chain dependency and uploads inside the clock.  step: the synthetic back-to-back step proof (rounds 1-2's headline; "
                         "profiling scripts); with ivc it is still measured and reported as step_micro")
    ap.add_argument("--chains", type=int, default=int(os.environ.get("VPBS_BENCH_CHAINS", "0")),
                    help="ivc workload: independent vPBS chains (own keys, context, witness plans, host threads) proven side by side per GPU.  "
                         "One chain leaves the GPU idle during its host phases; the metric is throughput, so the default (0 = auto) is 6 where "
                         "this rank's share of the host CPUs carries it (three chains per eight CPUs, measured), fewer on a small CPU "
                         "quota (the single-chain latency figure is reported next to it as ivc_single_chain)")
    ap.add_argument("--device-witness", type=int, default=int(os.environ.get("VPBS_BENCH_DEVICE_WITNESS", "-1")),
                    help="ivc workload: generate the early witness phases of this many steps at a time on the device "
                         "(vpbs_ivc_set_device_witness; the host keeps the late phase); 0 = host pipeline; -1 (default) = auto: on the device "
                         "where this rank's CPU share is too small to carry the host pipeline")
    ap.add_argument("--device-late", action="store_true", default=os.environ.get("VPBS_BENCH_DEVICE_LATE", "") not in ("", "0"),
                    help="with --device-witness: the late witness phase on the device too (the host generates no witness).  Automatic where this "
                         "rank has ONE CPU (measured with round 6's staged walk: 1 CPU 0.164 against 0.127 with the late phase on the host; 2 and 4 "
                         "CPUs equal, 0.173-0.175 / 0.185; from 12 CPUs on the host pipeline is faster); VPBS_BENCH_DEVICE_LATE=0 keeps it off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-step-micro", action="store_true", help="ivc workload: skip the synthetic step legs (step_micro, its batch, the parity check at full size)")
    ap.add_argument("--no-single-chain", action="store_true", help="ivc workload: skip the one-chain latency measurement")
    ap.add_argument("--no-step-circuit", action="store_true", help="skip the witness -> proof pipeline on the real step circuit")
    ap.add_argument("--no-whole-pbs", action="store_true", help="skip tools/prove_pbs.py (one whole vPBS, 730 step proofs, end to end)")
    ap.add_argument("--no-survey-size", action="store_true", help="skip the secondary degree-2^15 measurement (profiling runs)")
    ap.add_argument("--no-ivc", action="store_true", help="skip the FULL 730-step IVC chains in their own processes (tools/prove_ivc.py: one chain, three chains)")
    ap.add_argument("--no-batch128", action="store_true", help="skip the BASELINE config 3 leg (128 independent proofs through a pool of contexts)")
    ap.add_argument("--step-chains", type=int, default=1, help="step workload: independent synthetic chains proven concurrently per GPU")
    ap.add_argument("--batch-chains", type=int, default=4,
                    help="after the single-chain synthetic step, also time this many concurrent synthetic chains (1 GPU only)")
    ap.add_argument("--mode", choices=["replicas", "sharded", "sharded-replay"], default="replicas",
                    help="N > 1: 'replicas' = independent chains per GPU (weak scaling, default, no data-path collective); "
                         "'sharded' = ONE chain whose commitments are coset-sharded over the GPUs (strong scaling: per-step "
                         "latency; collectives: all-gather of cap hashes + one all-reduce of query records per step).  "
                         "'sharded-replay' (ONE GPU, not a bench line): per-rank compute time of the sharded step for worlds 2 / 4 / 8, each "
                         "rank alone on the device with its collectives answered from a recording (tools/sharded_replay.py)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None, help="force the HIP device ordinal (testing N > 1 on one GPU)")
    ap.add_argument("--detail", default=None, help="where the full result goes (default: bench_detail.json next to this script, and a copy under "
                                                   "gpurun_out/ where that directory exists); stdout carries the compact record only")
    ap.add_argument("--log-n", type=int, default=LOG_N, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.mode == "sharded-replay":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import sharded_replay
        print(json.dumps(sharded_replay.run(args.log_n, args.steps if args.steps <= 50 else 10, device=args.device or 0)))
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    # The circuit files the legs load are a BUILD product (__graft_entry__.build() -> tools/export_circuits.py; the product package only locates
    # them).  A checkout that was never built gets them here, by the harness running that build step in a process of its own -- not by the
    # product: before any device work, rank 0 only (the other ranks of a node find the files when their legs start, after the first barrier).
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        from vpbs_amd import circuit_file as _cf
        try:
            _cf.find_step_circuit(1024, 2, 4, 5, 728)
            _cf.find_cyclic_circuit(IVC_N, IVC_K, IVC_ELL, IVC_LOGB, IVC_NLWE, LOG_N)
            _cf.find_cyclic_circuit(2048, 2, 4, 5, 728, 17)
        except FileNotFoundError as e:
            import subprocess
            print("bench.py: %s -- running tools/export_circuits.py (the build step) first" % str(e)[:120], file=sys.stderr)
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "export_circuits.py")], stdout=subprocess.DEVNULL)
    if world != args.gpus:
        # an 8-GPU lease must not turn into an N = 1 line: --gpus is what the line will claim, the ranks are what runs
        print("bench.py: --gpus %d but %d rank(s) are running (WORLD_SIZE); start it as `python bench.py --gpus N` (it launches its own ranks) "
              "or under torch.distributed.run with --nproc-per-node N" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.device is not None:
        local_rank = args.device
    elif distributed and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):   # counting devices does not open one
        print("bench.py: rank %d: %d ranks on this node but %d device(s) visible; pass --device D to run all ranks on one device on purpose"
              % (rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    if distributed:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=vpbs_amd.sharding.group_timeout())
        else:
            dist.init_process_group(args.dist_backend, timeout=vpbs_amd.sharding.group_timeout())
    launch = launch_report(args, rank, local_rank, world, distributed)
    log_n = args.log_n
    secondary = rank == 0 and world == 1 and log_n == LOG_N
    # What a rank's share of the host CPUs carries (measured on one MI355X with the affinity mask as the share and 8 hardware queues,
    # tools/experiments/cpu_share.sh, vPBS proofs/s per GPU): with the early witness phases on the HOST 2 / 4 / 8 / 16 CPUs give 0.074 / 0.086 /
    # 0.144 / 0.163 (1 / 1 / 4 / 6 chains); with the early phases on the DEVICE in batches of 64 (the host keeps the late phase and one
    # hash-chain thread per chain) 0.123 / 0.159 / 0.163 / 0.164 (4 / 6 / 6 / 6 chains).  The device pipeline is the default wherever the
    # share is below 12 CPUs -- the ranks of a multi-GPU node on a small container -- and the host pipeline, equal there and two rounds
    # older, from 12 CPUs on.
    cpus = vpbs_amd.api.host_set_cpu_budget(0) // max(1, world)          # this rank's share of the CPUs the container may use
    dw_given = args.device_witness >= 0
    if args.device_witness < 0:
        # The batch (steps whose early phases run on the device at once) is half the timed steps, 8 .. 64: the timed region then holds two
        # whole batches -- what a chain in steady state runs per that many proofs.  (A batch of 64 around a window of 60 steps would leave the
        # early phases' device time outside the clock: 0.155 / 0.163 vPBS/s at 2 / 4 CPUs instead of the 0.145 / 0.152 of whole chains,
        # tools/experiments/bench_batch_few_cpus.sh.)  Whole chains (ivc_full_chains, tools/prove_ivc.py) use 64: ~5 % less device time per step.
        args.device_witness = max(8, min(64, args.steps // 2)) if cpus < 12 else 0
    if args.chains <= 0:
        # host pipeline: 3 / 4 / 5 / 6 / 8 / 10 chains per GPU = 8.89 / 8.58 / 8.51-8.65 / 8.42-8.52 / 8.60-8.68 / 9.5 ms per chained proof on
        # 16 CPUs with spinning waits (tools/experiments/chains_ab.sh)
        # device pipeline: eight chains at every share (a chain's late phase and the scatter of its values are serial with its proof, and with
        # few CPUs they are long: 4 CPUs 0.145-0.149 with six chains, 0.152 with eight or ten; 2 CPUs 0.140 either way: tools/experiments/ivc_matrix.sh --preset hw_queues_dw2)
        # host pipeline since the waits sleep (round 4): 6 / 7 / 8 / 9 / 10 / 12 chains = 8.17 / 8.02 / 7.99 / 8.21 / 8.13 / 8.19 ms per chained proof
        # on 16 CPUs (tools/experiments/ivc_matrix.sh --preset host_chains_16cpus): a chain per two CPUs, eight at most (one per hardware queue)
        args.chains = 8 if args.device_witness else max(1, min(8, cpus // 2))
        chains_why = "auto"
    else:
        chains_why = "--chains / VPBS_BENCH_CHAINS"
    if args.device_witness and cpus < 2 and os.environ.get("VPBS_BENCH_DEVICE_LATE", "") != "0":
        args.device_late = True
    launch["cpus_per_rank"] = cpus
    launch["pipeline"] = {
        "early_witness_phase": "device, %d steps per batch" % args.device_witness if args.device_witness else "host (a second thread per chain)",
        "chains_per_gpu": 1 if (distributed and args.mode == "sharded") else args.chains, "chains_chosen_by": chains_why,
        "hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
        "why": ("--device-witness / VPBS_BENCH_DEVICE_WITNESS given" if dw_given else
                "this rank's share of the host is %d CPU(s) (affinity mask and cgroup quota / ranks of the node): %s"
                % (cpus, "below 12 the host cannot carry the early witness phases of eight chains, so they run on the device in batches and the "
                         "host keeps the late phase (measured: 0.146-0.151 vPBS/s per GPU at 2 CPUs, 0.155-0.157 at 4; host pipeline there "
                         "0.074 / 0.086)" if args.device_witness else
                         "from 12 up the host pipeline is the faster one (0.166-0.172 against 0.161-0.163 on 16 CPUs), one chain per two CPUs, "
                         "eight at most"))}
    if distributed and args.mode == "sharded":
        launch["sharded_ceiling"] = sharded_ceiling(world)

    out, state = None, None
    if args.workload == "ivc":
        out = measure_ivc(args, rank, local_rank, world, distributed)
        if secondary and not args.no_single_chain and args.chains > 1:
            one = argparse.Namespace(**vars(args))
            one.chains = 1
            r1 = measure_ivc(one, rank, local_rank, world, distributed)
            # the dominant kernel's roofline is priced where its launches have the device to themselves: with several chains in flight a
            # leaf-hash launch shares the CUs with the other chains' kernels and its event time measures the sharing, not the kernel
            conc = out["roofline"]
            out["roofline"] = dict(r1["roofline"], measured_in="the ivc_single_chain leg of this run (one chain: every launch alone on the device; HIP "
                                                               "events on the prover's stream over its timed chained steps)",
                                   step_hbm_frac=conc.get("step_hbm_frac"), step_hbm_frac_single_chain=r1["roofline"].get("step_hbm_frac"),
                                   valu_budget=conc.get("valu_budget"), valu_budget_single_chain=r1["roofline"].get("valu_budget"),
                                   concurrent_chains={"chains": args.chains, "kernel_ms_per_step": conc["kernel_ms_per_step"], "launches": conc["launches"],
                                                      "shader_clock_mhz_in_kernel": conc["shader_clock_mhz_in_kernel"],
                                                      "what": "the same kernel's event time per step proof inside the headline region, where launches "
                                                              "of different chains overlap on the device"})
            out["ivc_single_chain"] = {"what": "ONE chain alone on the GPU, same clock placement: the latency of a chained step (the GPU waits "
                                               "for the chain's host phases)", "ms_per_step": r1["ms_per_step"], "steps": r1["steps"], "warmup": r1["warmup"],
                                       "vpbs_proofs_per_s": r1["value"], "seconds_per_vpbs_extrapolated": r1["ms_per_step"] * STEPS_PER_VPBS / 1e3,
                                       "chain_ms_per_step_split": r1["chain_ms_per_step_split"],
                                       "leaf_hash_ms_per_step": r1["roofline"]["kernel_ms_per_step"],
                                       "shader_clock_mhz_in_kernel": r1["roofline"]["shader_clock_mhz_in_kernel"]}
        if secondary and not args.no_step_micro:
            sargs = argparse.Namespace(**vars(args))
            sargs.steps, sargs.warmup = min(args.steps, 20), min(max(args.warmup, 1), 5)
            micro, state = measure_step(sargs, rank, local_rank, world, distributed, log_n)
            out["step_micro"] = {k: micro[k] for k in ("ms_per_step_proof", "step_proofs_per_s", "kernel_ms_one_step") if k in micro}
            out["step_micro"].update({"what": micro["config"]["workload"], "vpbs_proofs_per_s_by_step_rate": micro["value"],
                                      "steps": sargs.steps, "warmup": sargs.warmup,
                                      "leaf_hash_ms_per_step": micro["roofline"]["kernel_ms_per_step"],
                                      "int_issue_frac": micro["roofline"]["int_issue_frac"], "hbm_frac": micro["roofline"]["frac"],
                                      "shader_clock_mhz_in_kernel": micro["roofline"]["shader_clock_mhz_in_kernel"]})
            if "batch" in micro:
                out["step_micro"]["batch"] = micro["batch"]
    else:
        out, state = measure_step(args, rank, local_rank, world, distributed, log_n)

    if rank == 0:
        out.update(launch)
        if secondary and state is not None and not args.no_survey_size:
            out["survey_degree_2pow15"] = survey_size_leg(local_rank, args if args.workload == "step" else sargs, state)
        gpu_proof = state["gpu_proof"] if state is not None else None
        if state is not None:
            for ctx in state["ctxs"]:
                ctx.close()
            state["ctxs"] = []
            state["keep"] = []
            torch.cuda.empty_cache()
        if secondary and not args.no_batch128:
            try:
                out["batch_of_128"] = batch_of_128(local_rank)
            except Exception as e:   # a secondary figure must not take the headline line down with it
                out["batch_of_128"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if secondary and not args.no_step_circuit:
            try:
                out["step_circuit_pipeline"] = step_circuit_pipeline(local_rank)
            except Exception as e:   # a secondary figure must not take the headline line down with it
                out["step_circuit_pipeline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                out["step_circuit_pipeline"]["device_witness"] = step_circuit_device_pipeline(local_rank)
            except Exception as e:
                out["step_circuit_pipeline"]["device_witness"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        env = dict(os.environ, VPBS_PBS_DEVICE=str(local_rank), WORLD_SIZE="1", RANK="0")
        if secondary and not args.no_whole_pbs:
            # one whole vPBS end to end on the step circuit WITHOUT recursion (tools/prove_pbs.py) in its own process
            out["whole_pbs"] = subprocess_json([sys.executable, os.path.join(ROOT, "tools", "prove_pbs.py")], env, 600)
        if secondary and not args.no_ivc:
            # the complete object: ONE vPBS proof = base proof + all 730 chained proofs, verify_pbs on the last, decrypted (own process) ...
            out["ivc_chain"] = subprocess_json([sys.executable, os.path.join(ROOT, "tools", "prove_ivc.py")], env, 900)
            # ... and the SUSTAINED form of the headline: as many whole chains side by side as the headline ran (every chain all 730 steps, base
            # proof, verify_pbs and decryption included), same witness pipeline -- what the burst figure `value` should agree with
            fenv = dict(env, VPBS_IVC_CHAINS=str(args.chains))
            if args.device_witness:
                fenv["VPBS_IVC_DEVICE_WITNESS"] = "64"
            d = subprocess_json([sys.executable, os.path.join(ROOT, "tools", "prove_ivc.py")], fenv, 1200)
            if "error" in d:
                out["ivc_full_chains"] = d
            else:
                out["ivc_full_chains"] = {k: d[k] for k in ("what", "chains", "seconds", "vpbs_proofs_per_s", "ms_per_step", "ms_per_step_split",
                                                            "decrypted", "other_chains", "host") if k in d}
                sustained = d["vpbs_proofs_per_s"]
                out["sustained"] = {"vpbs_proofs_per_s": sustained, "chains": d["chains"], "seconds": d["seconds"],
                                    "burst_value": out["value"], "sustained_over_burst": sustained / out["value"],
                                    "what": "%d whole vPBS chains side by side, wall clock from the first base proof to the last chain's last proof "
                                            "(ivc_full_chains), against `value` = %d timed chained steps in steady state" % (d["chains"], args.steps),
                                    "why_they_differ": "the sustained figure also pays every chain's base proof and the ragged end (the chains do not "
                                                       "finish together: the last one proves alone at single-chain speed), and %.0f s of load give "
                                                       "the shared host more chances to interfere than a burst of %.1f s"
                                                       % (d["seconds"], out["ms_per_step"] * args.steps / 1e3)}
            # BASELINE config 5's ring: N = 2048 (src/ntt/params_2048.rs) -> the cyclic circuit at degree 2^17, LDE 2^20: chained steps of one chain
            d = subprocess_json([sys.executable, os.path.join(ROOT, "tools", "prove_ivc.py"), "2048", "728", "17", "24"], env, 900)
            out["ivc_chain_n2048"] = {k: d[k] for k in ("what", "step_proofs", "seconds", "ms_per_step", "ms_per_step_split", "proof_bytes",
                                                        "verify_last_proof_ms", "host") if k in d} if "error" not in d else d
        if world == 1 and not args.no_cpu_baseline and log_n == LOG_N:
            parity, out["cpu_baseline"] = cpu_baseline(gpu_proof)
            out["parity_checked_full_size"] = parity is not None
            out["parity_full_size"] = parity
        # The line is a record (numbers and short identifiers, < 4 kB: tools/bench_record.py); the legs, the per-kernel tables and every sentence
        # that explains them are the detail: bench_detail.json next to this script (and under gpurun_out/), a short summary on stderr.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_record
        line = bench_record.dumps(bench_record.compact_record(out))
        wrote = bench_record.write_detail(out, path=args.detail)
        print("bench.py: detail -> %s" % (", ".join(wrote) or "(not writable)"), file=sys.stderr)
        sys.stderr.flush()
        print(line, flush=True)
    elif state is not None:
        for ctx in state["ctxs"]:
            ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
