"""GPU tests (-m gpu): the reference's step circuit without the recursive verifier (build_step_circuit,
/root/reference/src/vtfhe/ivc_based_vpbs.rs:80-155; described by circuitgen/step_circuit.py) through the whole product: compiled witness
generation on the host (vpbs_witness_plan), the native TFHE data path on the device for the expected accumulators
(vpbs_pbs_accumulator_chain / vpbs_blind_rotate_step), step proofs on the device with the gate constraints of the six gate types the
circuit uses, the host verifier -- a verifiable PBS with the IVC hand-over (accumulator, counter, hash chains) done by the caller
instead of the in-circuit verifier."""
import os
import time

import numpy as np
import pytest

import oracle as orc
import step_circuit as sc
import step_oracle
import tfhe_oracle as T
import export_circuits
import vpbs_amd
from vpbs_amd import api

pytestmark = pytest.mark.gpu
P = api.P
DIGEST = [0xA, 0xB, 0xC, 0xD]
N_ROUTED = 80


@pytest.fixture(scope="module")
def ctx():
    c = vpbs_amd.Context(0, log_n_max=16)
    yield c
    c.close()


class Prover:
    """circuit data committed once (constants + sigmas), one witness plan, then witness -> proof -> verification per step"""

    def __init__(self, ctx, circ):
        self.ctx, self.circ, b = ctx, circ, circ.built
        self.sigma = b.circuit.sigma_values()
        self.n_constants = b.constants.shape[0]
        self.cs = ctx.commit_values(np.concatenate([b.constants, self.sigma]))
        self.ncols = [self.n_constants + N_ROUTED, 135, 20, 16]
        self.targets = ([t for p in circ.acc_init for t in p] + [t for p in circ.acc_in for t in p] + circ.ggsw_flat +
                        [circ.counter, circ.mask] + circ.bsk_hash_in + circ.lwe_hash_in)
        self.plan = b.circuit.witness_plan([b.pos(t) for t in self.targets])

    def witness(self, acc_init, acc_in, ggsw_flat, counter, mask, bsk_hash_in, lwe_hash_in, out=None):
        values = np.concatenate([np.asarray(acc_init, np.uint64).reshape(-1), np.asarray(acc_in, np.uint64).reshape(-1),
                                 np.asarray(ggsw_flat, np.uint64).reshape(-1), np.array([counter, mask], np.uint64),
                                 np.asarray(bsk_hash_in, np.uint64), np.asarray(lwe_hash_in, np.uint64)])
        return self.plan.run(values, out=out)

    def prove(self, wires):
        b = self.circ.built
        pis = self.circ.public_inputs(wires)
        si = self.ctx.make_step_inputs(b.log_n, wires, None, None, self.cs, DIGEST, pis, sigmas=self.sigma, n_routed=N_ROUTED,
                                       n_constants=self.n_constants, gates=b.gates)
        return self.ctx.prove_step(si), pis

    def verify(self, proof, pis):
        b = self.circ.built
        return api.verify_step(proof, self.cs.cap(), self.ncols, DIGEST, pis, b.log_n, check_permutation=True, n_constants=self.n_constants,
                               n_routed=N_ROUTED, gates=b.gates)

    def close(self):
        self.plan.free()
        self.cs.free()


def test_verifiable_pbs_every_step_proven(ctx):
    """src/main.rs:40-65 + verified_pbs (ivc_based_vpbs.rs:276-371) at N = 8, n = 3 with noise-free keys: n + 2 step proofs, each
    step's public inputs (accumulator, counter, bootstrapping-key and LWE hash chains) handed to the next step's witness by the
    caller; accumulators equal the device's native chain, hashes equal verify_hash_output's chain, every proof verifies, the
    key-switched output decrypts to the message."""
    rng = np.random.default_rng(77)
    log_N, K, ELL, LOGB, n, p = 3, 2, 8, 8, 3, 2
    ring = T.Ring(log_N)
    N = ring.n
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n, orc.negacyclic_params(log_N))
    pr = Prover(ctx, circ)
    s_to, s_lwe, s_glwe, bsk, ksk = T.pbs_setup(ring, rng, n, K, ELL, LOGB, p)
    delta = T.get_delta(2 * p)
    acc_init = np.array([[0] * N for _ in range(K - 1)] + [T.get_testv(ring, p, delta)], np.uint64)
    bsk_flat, ksk_flat = np.stack([T.flatten_ggsw(g) for g in bsk]), T.flatten_ggsw(ksk)
    for m in (0, 1):
        ct = T.lwe_encrypt(rng, s_lwe, delta * m % P)
        accs = ctx.pbs_accumulator_chain(acc_init, ct, bsk_flat, ksk_flat, K, ELL, LOGB)       # the native data path (device)
        dummy = np.zeros(K * ELL * K * N, np.uint64)                                            # Ggsw::dummy_ct() of step 0
        ggsws = [dummy] + list(bsk_flat) + [ksk_flat]
        masks = [int(ct[n])] + [int(v) for v in ct[:n]] + [0]
        acc_in, bsk_hash, lwe_hash = acc_init, np.zeros(4, np.uint64), np.zeros(4, np.uint64)
        for step in range(n + 2):
            wires = pr.witness(acc_init, acc_in, ggsws[step], step + 1, masks[step], bsk_hash, lwe_hash)
            proof, pis = pr.prove(wires)
            assert pr.verify(proof, pis)
            assert step_oracle.verify_step(proof, pr.cs.cap(), pr.ncols, DIGEST, pis, circ.built.log_n)
            acc_out = np.array(pis[K * N + 1:2 * K * N + 1], np.uint64).reshape(K, N)
            assert pis[:K * N] == [int(v) for v in acc_init.reshape(-1)] and pis[K * N] == step + 1
            assert (acc_out == accs[step]).all(), step
            bsk_hash, lwe_hash = np.array(pis[-8:-4], np.uint64), np.array(pis[-4:], np.uint64)
            want_bsk, _ = api.hash_chain(np.stack(ggsws[:step + 1]))
            want_lwe, _ = api.hash_chain(np.array(masks[:step + 1], np.uint64).reshape(-1, 1))
            assert (bsk_hash == want_bsk).all() and (lwe_hash == want_lwe).all()
            wrong = list(pis)
            wrong[K * N + 1] ^= 1                                                              # another accumulator
            assert not pr.verify(proof, wrong)
            acc_in = acc_out
        m_bar = T.glwe_decrypt(ring, s_to, [[int(v) for v in acc_in[q]] for q in range(K)], K)[0]
        assert round(m_bar / delta) % (2 * p) == m
    pr.close()


def test_step_circuit_at_the_papers_parameters(ctx):
    """N = 1024, k = 1, ELL = 4, LOGB = 5, n = 728 (src/main.rs:20-28): 38 312 gate rows -> degree 2^16, the degree of the reference's
    step circuit.  A CMUX step and the key-switch step: witness by the compiled plan, accumulator equal to the device's native step,
    proof on the device, verified on the host."""
    rng = np.random.default_rng(1024)
    N, K, ELL, LOGB, n = 1024, 2, 4, 5, 728
    t0 = time.perf_counter()
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n, orc.negacyclic_params(10))
    t_build = time.perf_counter() - t0
    assert circ.built.log_n == 16
    t0 = time.perf_counter()
    pr = Prover(ctx, circ)
    t_setup = time.perf_counter() - t0
    f = lambda *shape: rng.integers(0, P, size=shape, dtype=np.uint64)
    wires = np.zeros((135, circ.built.n), np.uint64)
    report = []
    for counter in (5, n + 2):
        acc_init, acc_in, ggsw, mask, h1, h2 = f(K, N), f(K, N), f(K * ELL * K * N), int(f(1)[0]), f(4), f(4)
        t0 = time.perf_counter()
        pr.witness(acc_init, acc_in, ggsw, counter, mask, h1, h2, out=wires)
        t_wit = time.perf_counter() - t0
        t0 = time.perf_counter()
        proof, pis = pr.prove(wires)
        t_prove = time.perf_counter() - t0
        want = ctx.blind_rotate_step(acc_in[None], [mask], ggsw, K, ELL, LOGB, last_step=counter == n + 2)[0]
        assert pis[K * N + 1:2 * K * N + 1] == [int(v) for v in want.reshape(-1)]
        assert pis[-8:-4] == [int(v) for v in orc.hash_no_pad(np.concatenate([h1, ggsw]))]
        ok, msg = circ.built.circuit.check_witness(wires, api.hash_no_pad(np.array(pis, np.uint64)))
        assert ok, msg
        assert pr.verify(proof, pis)
        wrong = list(pis)
        wrong[-1] ^= 1
        assert not pr.verify(proof, wrong)
        if counter == 5:                                 # one wrong advice wire of a BaseSumGate row: not a witness of the circuit
            row = circ.built.row_kinds.index("base_sum")
            bad = wires.copy()
            bad[7, row] ^= np.uint64(1)
            ok_bad, msg_bad = circ.built.circuit.check_witness(bad, api.hash_no_pad(np.array(pis, np.uint64)))
            assert not ok_bad and "row %d" % row in msg_bad
            proof_bad, _ = pr.prove(bad)
            assert not pr.verify(proof_bad, pis)
        report.append((counter, t_wit, t_prove))
    print("\nstep circuit N=1024: %d rows, builder %.1f s, setup (sigma, commit, plan) %.2f s; " % (circ.built.used_rows, t_build, t_setup) +
          "; ".join("counter %d: witness %.0f ms, proof (incl. H2D of the wires) %.1f ms" % (c, 1e3 * a, 1e3 * b) for c, a, b in report))
    pr.close()


@pytest.mark.parametrize("N", [8, 64])
def test_cxx_host_proves_an_exported_step_circuit(N, tmp_path):
    """examples/prove_step_circuit.cpp: the step circuit handed over as data (tools/export_step_circuit.py: what the Rust side would
    export after builder.build()) to a plain C++ host of the C ABI -- layout, sigma, witness plan, check, commit, proof, verification;
    its witness reproduces the exported public inputs."""
    import subprocess
    import sys
    import __graft_entry__ as entry
    sys.path.insert(0, entry.ROOT + "/tools")
    import export_step_circuit as ex
    path = str(tmp_path / "step.bin")
    circ, pis = ex.export(path, N=N)
    exe = entry.build_example("prove_step_circuit")
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "proof verified: 1; with a wrong public input: 0" in r.stdout
    assert "degree 2^%d" % circ.built.log_n in r.stdout and "%d public inputs" % len(pis) in r.stdout


def test_step_circuit_n2048():
    """BASELINE config 5's ring (N = 2048, the reference's params_2048 tables): 78 555 gate rows -> degree 2^17, 8 201 public inputs.
    One CMUX step: witness by the compiled plan, accumulator equal to the device's native step, proof on the device, verified."""
    rng = np.random.default_rng(2048)
    N, K, ELL, LOGB, n = 2048, 2, 4, 5, 728
    c17 = vpbs_amd.Context(0, log_n_max=17)
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n, api.ntt_params(11))
    assert circ.built.log_n == 17 and circ.built.used_rows == 78555
    pr = Prover(c17, circ)
    f = lambda *shape: rng.integers(0, P, size=shape, dtype=np.uint64)
    acc_init, acc_in, ggsw, mask, h1, h2 = f(K, N), f(K, N), f(K * ELL * K * N), int(f(1)[0]), f(4), f(4)
    wires = pr.witness(acc_init, acc_in, ggsw, 9, mask, h1, h2)
    proof, pis = pr.prove(wires)
    want = c17.blind_rotate_step(acc_in[None], [mask], ggsw, K, ELL, LOGB)[0]
    assert pis[K * N + 1:2 * K * N + 1] == [int(v) for v in want.reshape(-1)]
    assert pr.verify(proof, pis)
    wrong = list(pis)
    wrong[3] ^= 1
    assert not pr.verify(proof, wrong)
    pr.close()
    c17.close()


def _plan_and_values(circ, seeds, counters):
    b = circ.built
    N, K, ELL, LOGB, n_lwe = circ.shape
    targets = ([t for p in circ.acc_init for t in p] + [t for p in circ.acc_in for t in p] + circ.ggsw_flat + [circ.counter, circ.mask] +
               circ.bsk_hash_in + circ.lwe_hash_in)
    plan = b.circuit.witness_plan([b.pos(t) for t in targets])
    cols = []
    for seed, counter in zip(seeds, counters):
        rng = np.random.default_rng(seed)
        v = rng.integers(0, P, size=len(targets), dtype=np.uint64)
        v[len(targets) - 10] = counter
        cols.append(v)
    return plan, np.ascontiguousarray(np.stack(cols, axis=1))     # [n_preset][batch]


@pytest.mark.parametrize("N,batch", [(8, 1), (8, 5), (64, 3)])
def test_device_witness_matches_the_host_plan(ctx, N, batch):
    """vpbs_witness_device_*: the level schedule replayed on the device for a batch of PartialWitnesses (first / CMUX / key-switch
    steps mixed) gives, instance by instance, exactly the wires of the host plan; public inputs read back; a proof from the
    device-resident wires verifies."""
    import torch
    K, ELL, LOGB, n_lwe = 2, 4, 5, 6
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n_lwe, api.ntt_params(N.bit_length() - 1))
    b = circ.built
    counters = [1, 3, n_lwe + 2, 2, 5][:batch]
    plan, values = _plan_and_values(circ, range(100, 100 + batch), counters)
    dev = api.WitnessDevice(ctx, plan, max_batch=8)
    dev.run(values)
    d_wires = torch.zeros((135, b.n), dtype=torch.int64, device="cuda")
    pi_pos = [b.pos(t) for t in b.public_inputs]
    for i in range(batch):
        want = plan.run(values[:, i])
        dev.wires(i, d_wires.data_ptr())
        got = d_wires.cpu().numpy().view(np.uint64)
        assert (got == want).all(), (i, np.argwhere(got != want)[:5])
        assert [int(v) for v in dev.read(i, pi_pos)] == circ.public_inputs(want)
    # a second run with another batch size reuses the object (new graph)
    dev.run(values[:, :1])
    dev.wires(0, d_wires.data_ptr())
    assert (d_wires.cpu().numpy().view(np.uint64) == plan.run(values[:, 0])).all()
    # proof straight from the device-resident wires
    pr = Prover(ctx, circ)
    pis = [int(v) for v in dev.read(0, pi_pos)]
    d_sigma = torch.from_numpy(pr.sigma.view(np.int64)).cuda()
    si = ctx.make_step_inputs(b.log_n, d_wires.data_ptr(), None, None, pr.cs, DIGEST, pis, on_device=True, shapes=(135, 20, 16),
                              sigmas=d_sigma.data_ptr(), n_routed=N_ROUTED, n_constants=pr.n_constants, gates=b.gates)
    assert pr.verify(ctx.prove_step(si), pis)
    # a value error in one instance fails the run: the accumulator output preset to something else
    out_pos = b.pos(circ.acc_out[0][0])
    plan2 = b.circuit.witness_plan(plan_positions(plan) + [out_pos])
    dev2 = api.WitnessDevice(ctx, plan2, max_batch=4)
    good = np.concatenate([values[:, :2], np.array([[int(plan.run(values[:, i])[out_pos]) for i in range(2)]], np.uint64)]) if batch >= 2 else None
    if good is not None:
        dev2.run(good)
        bad = good.copy()
        bad[-1, 1] ^= np.uint64(1)
        with pytest.raises(api.VpbsError, match="set twice"):
            dev2.run(bad)
    dev2.free(); dev.free(); pr.close(); plan.free(); plan2.free()


def plan_positions(plan):
    n = plan.circuit.n
    return [divmod(int(p), n) for p in plan.positions]


@pytest.mark.parametrize("world", [1, 2])
def test_whole_pbs_tool(world):
    """tools/prove_pbs.py on a short chain (n = 14: 16 steps) at the paper's ring dimension: every step witness-generated on the device,
    proven, verified, public inputs equal to the native chains; with two ranks (both on this box's one GPU, gloo) the steps of the one
    PBS are split between the ranks."""
    import json
    import subprocess
    import sys
    import __graft_entry__ as entry
    tool = entry.ROOT + "/tools/prove_pbs.py"
    export_circuits.ensure_step_circuit(1024, 2, 4, 5, 14)   # the tool loads circuit files, it does not make them
    env = dict(os.environ, VPBS_PBS_BACKEND="gloo", VPBS_PBS_DEVICE="0")
    if world == 1:
        cmd = [sys.executable, tool, "14", "4", "2"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
               "--master-port", "29561", tool, "14", "4", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["step_proofs"] == 16 and d["n_gpus"] == world and "all 16 proofs verified" in d["checks"]


def test_device_witness_at_paper_parameters_against_independent_evaluations(ctx):
    """Row a14 at full size: the device witness generator on a batch of 73 consecutive steps of a REAL PBS at the paper's parameters
    (N = 1024, K = 2, ELL = 4, LOGB = 5, n = 728; seeded keys with the paper's noise, vpbs_keygen) -- checked against evaluations that
    share no code with it:
      * every gathered wire matrix of a sample satisfies all gate constraints and all copy constraints (vpbs_check_witness: host
        re-evaluation of the 38 312 gate rows from the wires alone);
      * the accumulator public inputs equal the Python big-int restatement of the step (tests/tfhe_oracle.py, reference semantics
        ivc_based_vpbs.rs:99-125) applied to the previous accumulator;
      * counter and chain-hash public inputs equal the native sponge (hash_no_pad of previous hash || key material, :126-143);
      * and the wires equal the host plan's."""
    import torch
    import tfhe_oracle as T
    from vpbs_amd import circuit_file
    N, K, ELL, LOGB, n_lwe, batch = 1024, 2, 4, 5, 728, 73
    d = circuit_file.load(export_circuits.ensure_step_circuit(N, K, ELL, LOGB, n_lwe))
    keys = ctx.keygen(N, K, ELL, LOGB, n_lwe, 0xA14, 4.99027217501041e-8, 1.17021618159313e-5)
    testv, delta = api.testv(N, 2)
    ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta % P)
    acc_init = np.concatenate([np.zeros((K - 1, N), np.uint64), testv.reshape(1, N)])
    accs = ctx.pbs_accumulator_chain(acc_init, ct, keys["bsk"], keys["ksk"], K, ELL, LOGB)
    ggsw_len = K * ELL * K * N
    ggsws = lambda s: np.zeros(ggsw_len, np.uint64) if s == 0 else keys["bsk"][s - 1]
    masks = [int(ct[n_lwe])] + [int(v) for v in ct[:n_lwe]]
    bsk_h, lwe_h = [np.zeros(4, np.uint64)], [np.zeros(4, np.uint64)]
    for s in range(batch):
        bsk_h.append(api.hash_no_pad(np.concatenate([bsk_h[-1], ggsws(s)])))
        lwe_h.append(api.hash_no_pad(np.concatenate([lwe_h[-1], np.array([masks[s]], np.uint64)])))
    vals = np.zeros((len(d.preset_pos), batch), np.uint64)
    for s in range(batch):
        acc_in = acc_init if s == 0 else accs[s - 1]
        vals[:, s] = np.concatenate([acc_init.reshape(-1), acc_in.reshape(-1), ggsws(s), np.array([s + 1, masks[s]], np.uint64), bsk_h[s], lwe_h[s]])
    plan = d.circuit.witness_plan(d.preset_pos)
    dev = api.WitnessDevice(ctx, plan, max_batch=batch)
    dev.run(vals)
    d_wires = torch.zeros((135, d.n), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ring = T.Ring(10)
    for s in (0, 1, 36, 72):
        dev.wires(s, d_wires.data_ptr())
        w = d_wires.cpu().numpy().view(np.uint64)
        pis = np.array([w[c][r] for c, r in d.pi_pos], np.uint64)
        assert (dev.read(s, d.pi_pos) == pis).all()
        ok, msg = d.circuit.check_witness(w, api.hash_no_pad(pis))
        assert ok, (s, msg)
        # public inputs in the reference's order (ivc_based_vpbs.rs:196-207): acc_init, counter, accumulator, the two chain hashes
        assert (pis[:K * N] == acc_init.reshape(-1)).all() and int(pis[K * N]) == s + 1
        acc_in = acc_init if s == 0 else accs[s - 1]
        ggsw_hat = [[[[int(v) for v in ggsws(s).reshape(K, ELL, K, N)[p, l, r]] for r in range(K)] for l in range(ELL)] for p in range(K)]
        want = T.step(ring, [[int(v) for v in poly] for poly in acc_in], masks[s], ggsw_hat, K, ELL, LOGB, first_step=(s == 0))
        assert [int(v) for v in pis[K * N + 1:2 * K * N + 1]] == [c for poly in want for c in poly], s
        assert (pis[-8:-4] == bsk_h[s + 1]).all() and (pis[-4:] == lwe_h[s + 1]).all()
        if s in (1, 72):
            assert (plan.run(vals[:, s]) == w).all()
    dev.free()
    plan.free()


@pytest.mark.parametrize("args,expect_steps", [(["8", "6", "13"], 8), (["1024", "728", "16", "5"], 5), (["2048", "728", "17", "3"], 3)])
def test_ivc_chain_tool(args, expect_steps):
    """tools/prove_ivc.py: the reference's IVC (ivc_based_vpbs.rs:159-386) on the GPU -- every step proof of the CYCLIC circuit verifies its
    predecessor in circuit; the last proof alone is verified (after a byte round trip) and carries test vector, counter, verifier data, the
    native accumulator and both native chain hashes.  N = 8: the whole chain of a PBS with n = 6 (BASELINE config 1's ring), decrypting to
    the message; N = 1024: the first five steps of the paper-parameter chain (degree 2^16, 4173 public inputs; BASELINE config 4's circuit);
    N = 2048 (BASELINE config 5's ring, the reference's params_2048 tables): 85 413 gate rows -> degree 2^17, 8269 public inputs."""
    import json
    import subprocess
    import sys
    import __graft_entry__ as entry
    export_circuits.ensure_cyclic_circuit(int(args[0]), 2, 4, 5, int(args[1]), int(args[2]))   # the tool loads circuit files, it does not make them
    r = subprocess.run([sys.executable, entry.ROOT + "/tools/prove_ivc.py"] + args, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["step_proofs"] == expect_steps
    if args[0] == "8":
        assert d["decrypted"] == d["message"] == 1
    elif args[0] == "1024":
        assert 150_000 < d["proof_bytes"] < 210_000                 # the paper's "~200 kB" (ivc_based_vpbs.rs:488)
    print(d["ms_per_step_split"], d["seconds"])


@pytest.mark.parametrize("N,n_lwe,log_n,steps,device_witness", [(8, 6, 13, 8, 0), (1024, 728, 16, 4, 0), (8, 6, 13, 8, 3), (1024, 728, 16, 5, 2)])
def test_cxx_host_proves_an_ivc_chain(N, n_lwe, log_n, steps, device_witness):
    """examples/prove_ivc.cpp: the IVC chain driven by a plain C++ host of the C ABI (no Python, no torch in the process) from the exported
    cyclic + dummy circuit files: split witness plan with the early phase on its own thread, pinned wires, the final proof alone verified
    after a byte round trip, chain hashes, and -- for the whole chain at N = 8 -- decryption to the message."""
    import subprocess
    import __graft_entry__ as entry
    from vpbs_amd import circuit_file
    cyc, dum = export_circuits.ensure_cyclic_circuit(N, 2, 4, 5, n_lwe, log_n)
    exe = entry.build_example("prove_ivc")
    env = dict(os.environ, VPBS_IVC_DEVICE_WITNESS=str(device_witness)) if device_witness else dict(os.environ)   # the other witness pipeline
    r = subprocess.run([exe, cyc, dum, str(steps)], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "IVC chain: %d of %d step proofs" % (steps, n_lwe + 2) in r.stdout and "verified: 1" in r.stdout
    if steps == n_lwe + 2:
        assert "decrypted 1 (message 1)" in r.stdout
    print(r.stdout.strip())


def test_ivc_chain_bit_identical_to_the_cpu_oracle_chain():
    """The N = 8, n = 1 chain of tests/test_cyclic_cpu.py (same keys, ciphertext, test vector) through vpbs_ivc_prove_pbs on the GPU: the
    serialised last proof has the bytes the CPU oracle's chain froze in tests/golden/ivc_chain_n8.json -- three chained proofs of the cyclic
    circuit, witness generation by the split plans, every prover stage and the transcript, bit for bit."""
    import hashlib
    import json
    from test_cyclic_cpu import GOLDEN_CHAIN, n8_chain_inputs
    from vpbs_amd import circuit_file
    N, K, ELL, LOGB, n_lwe, log_n = 8, 2, 4, 5, 1, 13
    ring, (s_to, s_lwe, s_glwe, bsk, ksk), delta, testv, ct = n8_chain_inputs()
    cyc, dum = (circuit_file.load(p) for p in export_circuits.ensure_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n))
    c = vpbs_amd.Context(0, log_n_max=16)
    ivc = api.Ivc(c, cyc, dum, N, K, K * ELL * K * N)
    bsk_flat, ksk_flat = np.stack([T.flatten_ggsw(g) for g in bsk]), T.flatten_ggsw(ksk)
    blob, _ = ivc.prove_pbs(testv, ct, bsk_flat, ksk_flat)
    frozen = json.load(open(GOLDEN_CHAIN))
    assert (len(blob), hashlib.sha256(blob).hexdigest()) == (frozen["bytes"], frozen["sha256"])
    vk, _ = ivc.verifier_data()
    # the bootstrapped ciphertext the verifier holds: the native accumulator chain's last element (the statement the proof is bound to)
    acc_init = np.concatenate([np.zeros((K - 1, N), np.uint64), np.asarray(testv, np.uint64).reshape(1, N)])
    out_ct = c.pbs_accumulator_chain(acc_init, ct, bsk_flat, ksk_flat, K, ELL, LOGB)[-1]
    ok, why = api.verify_pbs(blob, vk[4:].reshape(-1, 4), [cyc.n_constants + 80, 135, 20, 16], vk[:4], log_n, cyc.n_constants, 80, cyc.gates, N, K,
                             testv, ct, bsk_flat, ksk_flat, out_ct)
    assert ok, why
    with pytest.raises(api.VpbsError):   # no verdict without the ciphertext (ivc_based_vpbs.rs:440-442)
        api.verify_pbs(blob, vk[4:].reshape(-1, 4), [cyc.n_constants + 80, 135, 20, 16], vk[:4], log_n, cyc.n_constants, 80, cyc.gates, N, K,
                       testv, ct, bsk_flat, ksk_flat, None)
    ivc.free()
    c.close()


def test_ivc_driver_through_the_python_binding():
    """vpbs_ivc_create checks the PartialWitness / public-input layout against the parameters; vpbs_ivc_prove_pbs twice on one object (two
    PBS with the same keys, different ciphertexts), each proof accepted by vpbs_verify_pbs for its own ciphertext only"""
    from vpbs_amd import circuit_file
    N, K, ELL, LOGB, n_lwe, log_n = 8, 2, 4, 5, 6, 13
    cyc, dum = (circuit_file.load(p) for p in export_circuits.ensure_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n))
    c = vpbs_amd.Context(0, log_n_max=16)
    g = K * ELL * K * N
    with pytest.raises(api.VpbsError, match="not a cyclic step circuit"):
        api.Ivc(c, cyc, dum, N, K, g + 1)
    with pytest.raises(api.VpbsError, match="not a cyclic step circuit"):
        api.Ivc(c, dum, cyc, N, K, g)
    ivc = api.Ivc(c, cyc, dum, N, K, g)
    keys = c.keygen(N, K, ELL, LOGB, n_lwe, 77, 4.99027217501041e-8, 1.17021618159313e-5)
    testv, delta = api.testv(N, 2)
    ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta % P)
    blob, t = ivc.prove_pbs(testv, ct, keys["bsk"], keys["ksk"])
    vk, _ = ivc.verifier_data()
    acc_init = np.concatenate([np.zeros((K - 1, N), np.uint64), np.asarray(testv, np.uint64).reshape(1, N)])
    out_of = lambda c_: c.pbs_accumulator_chain(acc_init, c_, keys["bsk"], keys["ksk"], K, ELL, LOGB)[-1]   # the bootstrapped ciphertext
    ok, why = api.verify_pbs(blob, vk[4:].reshape(-1, 4), [cyc.n_constants + 80, 135, 20, 16], vk[:4], log_n, cyc.n_constants, 80, cyc.gates, N, K,
                             testv, ct, keys["bsk"], keys["ksk"], out_of(ct))
    assert ok, why
    assert t["steps"] == n_lwe + 2
    ct2 = api.lwe_encrypt(keys["params"], keys["s_lwe"], 0, nonce=1)
    # the progress hook: 0 after the base proof, then once per chained step, on the proving thread
    seen = []
    ivc.on_step(seen.append)
    blob2, _ = ivc.prove_pbs(testv, ct2, keys["bsk"], keys["ksk"])
    assert seen == list(range(n_lwe + 3))
    ivc.on_step(None)
    vp = lambda b, c_, o_: api.verify_pbs(b, vk[4:].reshape(-1, 4), [cyc.n_constants + 80, 135, 20, 16], vk[:4], log_n, cyc.n_constants, 80, cyc.gates,
                                          N, K, testv, c_, keys["bsk"], keys["ksk"], o_)
    assert vp(blob2, ct2, out_of(ct2))[0] and vp(blob2, ct, out_of(ct2)) == (False, "the LWE hash chain does not match")
    assert vp(blob2, ct2, out_of(ct)) == (False, "the output ciphertext is not the proof's accumulator") and not vp(blob, ct2, out_of(ct2))[0]
    ivc.free()
    # a context of another FriConfig shape cannot carry the chain (the in-circuit verifier is built for rate 1/8, cap height 4): rejected at
    # creation instead of overflowing the cap buffer or producing a verifier key that fails later in circuit
    for rate_bits, cap_height in ((3, 5), (3, 3), (2, 4)):
        c2 = vpbs_amd.Context(0, log_n_max=16, rate_bits=rate_bits, cap_height=cap_height)
        with pytest.raises(api.VpbsError, match="rate_bits = 3 and cap_height = 4"):
            api.Ivc(c2, cyc, dum, N, K, g)
        c2.close()
    c.close()


def test_ivc_driver_with_the_native_rccl_communicator_on_one_rank():
    """vpbs_ivc_create with a communicator made by vpbs_comm_rccl_create (RCCL bound with dlopen; a world of one rank is all a one-GPU box
    allows): sharded constants / sigmas commitment, every step through vpbs_prove_step_sharded with ncclAllGather / ncclAllReduce on the
    prover's stream -- and the last proof still has the bytes of the CPU oracle's chain"""
    import hashlib
    import json
    from test_cyclic_cpu import GOLDEN_CHAIN, n8_chain_inputs
    from vpbs_amd import circuit_file, sharding
    if not api.lib().vpbs_rccl_available():
        pytest.skip("librccl.so is not loadable here")
    N, K, ELL, LOGB, n_lwe, log_n = 8, 2, 4, 5, 1, 13
    ring, (s_to, s_lwe, s_glwe, bsk, ksk), delta, testv, ct = n8_chain_inputs()
    cyc, dum = (circuit_file.load(p) for p in export_circuits.ensure_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n))
    c = vpbs_amd.Context(0, log_n_max=16)
    comm = sharding.make_comm_rccl(c, stage_words=2 << (log_n + 3))
    ivc = api.Ivc(c, cyc, dum, N, K, K * ELL * K * N, comm)
    blob, _ = ivc.prove_pbs(testv, ct, np.stack([T.flatten_ggsw(g) for g in bsk]), T.flatten_ggsw(ksk))
    frozen = json.load(open(GOLDEN_CHAIN))
    assert (len(blob), hashlib.sha256(blob).hexdigest()) == (frozen["bytes"], frozen["sha256"])
    ivc.free()
    sharding.free_comm_rccl(comm)
    c.close()


def test_ivc_chain_tool_with_the_loop_spelled_out_in_python():
    """VPBS_IVC_DRIVER=python: the same chain driven call by call over the C ABI (run_early / upload_bg / run_late / upload_rows / prove_step)
    instead of vpbs_ivc_prove_pbs -- the form a host that wants its own pipeline would write"""
    import json
    import subprocess
    import sys
    import __graft_entry__ as entry
    r = subprocess.run([sys.executable, entry.ROOT + "/tools/prove_ivc.py", "8", "6", "13"], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, VPBS_IVC_DRIVER="python"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["driver"].startswith("the loop of tools/prove_ivc.py") and d["step_proofs"] == 8 and d["decrypted"] == d["message"] == 1


def test_ivc_chain_tool_two_chains_side_by_side():
    """VPBS_IVC_CHAINS=2: two independent PBS (own seed, message, context and witness plans) chained concurrently on the one GPU; each
    final proof passes verify_pbs and decrypts to its own message"""
    import json
    import subprocess
    import sys
    import __graft_entry__ as entry
    r = subprocess.run([sys.executable, entry.ROOT + "/tools/prove_ivc.py", "8", "6", "13"], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, VPBS_IVC_CHAINS="2"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["chains"] == 2 and d["step_proofs"] == 8 and d["decrypted"] == d["message"] == 1
    assert [c["decrypted"] for c in d["other_chains"]] == [c["message"] for c in d["other_chains"]] == [0]


@pytest.mark.parametrize("device_witness", [0, 3])
def test_ivc_chain_tool_sharded_over_two_ranks(device_witness):
    """BASELINE config 4's mechanism on the one GPU of the test box: the IVC chain with every step proof coset-sharded over two ranks (gloo,
    callback communicator; both ranks on device 0) -- same final proof checks as the single-rank chain, decrypting to the message; with the
    host witness pipeline and with the early phases on the device (every rank generates the identical witnesses either way)"""
    import json
    import subprocess
    import sys
    import __graft_entry__ as entry
    env = dict(os.environ, VPBS_PBS_BACKEND="gloo", VPBS_PBS_DEVICE="0", VPBS_IVC_DEVICE_WITNESS=str(device_witness))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29571",
           entry.ROOT + "/tools/prove_ivc.py", "8", "6", "13"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["step_proofs"] == 8 and d["decrypted"] == d["message"] == 1


def _bench_line_and_detail(stdout, detail_path):
    """the ONE compact record on stdout (< 4 kB: VERDICT r05 next 1) and the full result bench.py wrote to --detail"""
    import json
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, stdout[-2000:]
    line, d = json.loads(lines[0]), json.load(open(detail_path))
    assert line["detail"] == "bench_detail.json" and abs(line["value"] - d["value"]) <= 1e-5 * d["value"] and line["n_gpus"] == d["n_gpus"]
    assert line["metric"].startswith("vPBS proofs/sec at N=1024") and line["roofline"]["bound"] == "hbm" and line["roofline"]["kernel"] == "leaf_hash_kernel"
    return line, d


def test_bench_with_the_early_witness_phases_on_the_device(tmp_path):
    """bench.py --device-witness: the headline workload through the device pipeline (what a rank with a small CPU share runs by default);
    the same contract line, every chain's last proof checked after the clock"""
    import subprocess
    import sys
    import __graft_entry__ as entry
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, entry.ROOT + "/bench.py", "--steps", "10", "--warmup", "3", "--chains", "2", "--device-witness", "4",
                        "--no-single-chain", "--no-step-micro", "--no-cpu-baseline", "--no-survey-size", "--no-step-circuit", "--no-batch128",
                        "--no-whole-pbs", "--no-ivc", "--detail", detail], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line, d = _bench_line_and_detail(r.stdout, detail)
    assert line["steps"] == 10 and line["config"]["chains_per_gpu"] == 2 and line["config"]["witness"] == "device"
    assert d["steps"] == 10 and d["config"]["chains_per_gpu"] == 2 and d["config"]["early_witness_phase"].startswith("on the device, 4 steps")
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"] / 730) < 1e-9 and d["chain_checks"]["proof_bytes"] == 192716
    # the window sits in the pipeline's steady state: two batches of run-in before the warm-up, two batches of tail after the clock
    assert d["config"]["run_in_steps"] == 8 and d["config"]["tail_steps"] == 8 and "ON THE DEVICE" in d["config"]["stages"]


def test_bench_contract_with_two_ranks_sharing_the_gpu(tmp_path):
    """The driver's multi-GPU launch of bench.py (torch.distributed.run, one rank per GPU, replicas: independent chains per rank, no
    data-path collective) with both ranks on the one device of the test box (--device 0, gloo for the barriers): ONE JSON line from rank 0,
    n_gpus = 2, exactly --steps timed chained steps, value = the chains of BOTH ranks over the slower rank's time, every chain's last proof
    checked after the clock."""
    import json
    import subprocess
    import sys
    import __graft_entry__ as entry
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", entry.ROOT + "/bench.py", "--gpus", "2", "--steps", "12", "--warmup", "2", "--device", "0",
                        "--dist-backend", "gloo", "--chains", "1", "--device-witness", "0", "--detail", str(tmp_path / "detail.json")],
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line, d = _bench_line_and_detail(r.stdout, str(tmp_path / "detail.json"))
    assert (line["n_gpus"], line["steps"], line["warmup"], line["scaling"], line["vs_baseline"]) == (2, 12, 2, "weak", None)
    assert line["config"]["parallelism"] == "replicas" and "cpu_baseline" not in line and line["rccl"]["ranks"] == 2
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["vs_baseline"]) == (2, 12, 2, "weak", None)
    assert d["config"]["chains_per_gpu"] == 1 and "vpbs_ivc_prove_pbs" in d["config"]["workload"]
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"] / 730) < 1e-9 * d["value"] + 1e-12       # both ranks' chains over the MAX time
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["int_valu_issue"]["bound"] == "int-valu-issue" and d["chain_checks"]["proof_bytes"] == 192716
    assert "cpu_baseline" not in d                                                              # N = 1 only


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher around it: the parent starts torch.distributed.run as a child process
    and relays rank 0's line -- n_gpus = 2, the communication library's own count of the ranks, the CPUs per rank and the pipeline chosen.
    Both ranks on the one device of the test box (--device 0), gloo for the process group."""
    import json
    import subprocess
    import sys
    import __graft_entry__ as entry
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, entry.ROOT + "/bench.py", "--gpus", "2", "--steps", "12", "--warmup", "2", "--device", "0",
                        "--dist-backend", "gloo", "--chains", "1", "--device-witness", "0", "--detail", str(tmp_path / "detail.json")],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE line on stdout, everything else went to stderr
    line, d = _bench_line_and_detail(r.stdout, str(tmp_path / "detail.json"))
    assert (line["n_gpus"], line["steps"], line["warmup"], line["scaling"]) == (2, 12, 2, "weak") and line["rccl"] == {"ranks": 2, "version": line["rccl"]["version"], "backend": "gloo"}
    assert line["launched_by"].startswith("bench.py itself") and line["cpus_per_rank"] >= 1 and line["config"]["witness"] == "host"
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"]) == (2, 12, 2, "weak")
    assert d["rccl"]["ranks"] == 2 and d["rccl"]["backend"] == "gloo" and len(d["rccl"]["devices"]) == 2
    assert d["launched_by"].startswith("bench.py itself") and d["cpus_per_rank"] >= 1
    assert d["pipeline"]["chains_per_gpu"] == 1 and d["pipeline"]["early_witness_phase"].startswith("host")
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"] / 730) < 1e-9 * d["value"] + 1e-12
    # more ranks than devices without --device: refused before any rank starts (the box shows one GPU)
    import torch
    n_dev = torch.cuda.device_count()
    r = subprocess.run([sys.executable, entry.ROOT + "/bench.py", "--gpus", str(2 * n_dev), "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and r.stdout.strip() == "" and "visible" in r.stderr


def test_graft_entry_smoke_runs():
    """__graft_entry__.smoke() -- what the driver runs on the GPU box before the bench -- in its own process (it opens its own context)"""
    import subprocess
    import sys
    import __graft_entry__ as entry
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=entry.ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("smoke ok") >= 2


@pytest.mark.parametrize("N,n_lwe,log_n,batch", [(8, 6, 13, 3), (1024, 728, 16, 2)])
def test_device_early_phase_of_the_cyclic_circuit(ctx, N, n_lwe, log_n, batch):
    """vpbs_witness_device_create_early: the EARLY phase of the split cyclic-circuit plan (everything that does not need the previous proof)
    for a batch of steps on the device.  Per instance: the gathered matrix equals the host's early matrix (late positions zero), and the
    values read back for the host's late phase (vpbs_witness_device_read_late_inputs) are the host early phase's -- so a late phase seeded
    with them produces the same packed late values.  The late presets' rows of the batch are garbage on purpose."""
    import torch
    from vpbs_amd import circuit_file
    K, ELL, LOGB = 2, 4, 5
    cyc, _ = (circuit_file.load(p) for p in export_circuits.ensure_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n))
    W, n_pi = cyc.meta["proof_words"], len(cyc.pi_pos)
    plan = cyc.circuit.witness_plan(cyc.preset_pos)
    late = np.zeros(len(cyc.preset_pos), np.uint8)
    late[:W] = 1
    plan.split(late)
    rng = np.random.default_rng(N)
    fe = lambda k: rng.integers(0, int(P), size=k, dtype=np.uint64)
    g = K * ELL * K * N
    own_vk, dummy_vk, dummy_proof = fe(68), fe(68), fe(W)
    cols = []
    for i in range(batch):
        inner_pis = fe(n_pi)
        inner_pis[-68:] = own_vk                                             # connected to the circuit's own verifier data
        cols.append(np.concatenate([fe(W), inner_pis, np.array([i % 2], np.uint64), fe(g), fe(1), own_vk, dummy_vk, dummy_proof,
                                    np.zeros(n_pi, np.uint64)]))
    values = np.ascontiguousarray(np.stack(cols, axis=1))
    assert values.shape == (len(cyc.preset_pos), batch)
    dev = api.WitnessDevice(ctx, plan, max_batch=4, early=True)
    dev.run(values)
    d_wires = torch.zeros((135, cyc.n), dtype=torch.int64, device="cuda")
    lin, lpos = plan.late_input_positions(), plan.late_positions()
    for i in range(batch):
        host = np.zeros((135, cyc.n), np.uint64)
        state = plan.run_early(values[:, i], host)
        dev.wires(i, d_wires.data_ptr())
        got = d_wires.cpu().numpy().view(np.uint64)
        assert (got == host).all(), (i, np.argwhere(got != host)[:5])
        assert (got.reshape(-1)[lpos] == 0).all()
        seed = dev.read_late_inputs(i)
        assert (seed == host.reshape(-1)[lin]).all()
        api.lib().vpbs_witness_state_free(state)
    # a late phase on the device cannot succeed on garbage proof words: the call fails and says which class caught it
    with pytest.raises(api.VpbsError, match="device witness generation"):
        dev.run_late(0, values[:, 0])
    dev.free()
    plan.free()


def test_ivc_chain_with_the_early_phases_on_the_device():
    """vpbs_ivc_set_device_witness: the chain with the early witness phases generated on the device in batches (the host keeps the late
    phase): the N = 8 chain ends in the frozen bytes of the CPU oracle's chain -- with a batch smaller than the chain, equal to it and larger
    -- and going back to the host pipeline on the same object gives them again; at the paper's parameters a prefix of the chain (two
    batches) passes the checks of verify_pbs's prefix form."""
    import hashlib
    import json
    import sys
    import __graft_entry__ as entry
    from test_cyclic_cpu import GOLDEN_CHAIN, n8_chain_inputs
    from vpbs_amd import circuit_file
    N, K, ELL, LOGB, n_lwe, log_n = 8, 2, 4, 5, 1, 13
    ring, (s_to, s_lwe, s_glwe, bsk, ksk), delta, testv, ct = n8_chain_inputs()
    cyc, dum = (circuit_file.load(p) for p in export_circuits.ensure_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n))
    c = vpbs_amd.Context(0, log_n_max=16)
    ivc = api.Ivc(c, cyc, dum, N, K, K * ELL * K * N)
    frozen = json.load(open(GOLDEN_CHAIN))
    bsk_flat, ksk_flat = np.stack([T.flatten_ggsw(g) for g in bsk]), T.flatten_ggsw(ksk)
    for batch, late in ((2, False), (3, False), (8, False), (2, True), (8, True), (0, False)):
        ivc.set_device_witness(ELL, LOGB, batch, late)
        blob, t = ivc.prove_pbs(testv, ct, bsk_flat, ksk_flat)
        assert (len(blob), hashlib.sha256(blob).hexdigest()) == (frozen["bytes"], frozen["sha256"]), (batch, late)
    with pytest.raises(api.VpbsError):
        ivc.set_device_witness(ELL + 1, LOGB, 4)                      # does not fit the GGSW length of the circuit
    ivc.free()
    c.close()
    # paper parameters: 7 chained steps in batches of 3 through the tool's checks (accumulator, counter, both hash-chain prefixes, verifier data)
    sys.path.insert(0, entry.ROOT + "/tools")
    import prove_ivc
    N, n_lwe, log_n, steps = 1024, 728, 16, 7
    cyc, dum = (circuit_file.load(p) for p in export_circuits.ensure_cyclic_circuit(N, K, ELL, LOGB, n_lwe, log_n))
    c = vpbs_amd.Context(0, log_n_max=16)
    ivc = api.Ivc(c, cyc, dum, N, K, K * ELL * K * N)
    keys = c.keygen(N, K, ELL, LOGB, n_lwe, 5, 4.99027217501041e-8, 1.17021618159313e-5)
    tv, delta = api.testv(N, 2)
    ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta % P)
    vk, _ = ivc.verifier_data()
    blobs = []
    for late in (False, True):
        ivc.set_device_witness(ELL, LOGB, 3, late)
        blob, t = ivc.prove_pbs(tv, ct, keys["bsk"], keys["ksk"], steps)
        prove_ivc.check_chain(c, cyc, vk, blob, keys, tv, delta, ct, N, n_lwe, log_n, steps, 1)
        assert t["steps"] == steps and t["early_witness_ms"] > 0
        blobs.append(blob)
    assert blobs[0] == blobs[1]                                       # the late phase on the host or on the device: the same proof
    ivc.free()
    c.close()
